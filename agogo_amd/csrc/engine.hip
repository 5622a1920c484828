// placeholder until the device MCTS engine lands (next commit)
#include "common.hpp"
#define NYI(name) { agz::set_error(name ": not built yet"); return AGZ_E_UNSUPPORTED; }
extern "C" {
int agz_arena_create(agz_ctx*, const agz_game_conf*, const agz_mcts_conf*, int, uint64_t, int, agz_arena**) NYI("agz_arena_create")
void agz_arena_destroy(agz_arena*) {}
int agz_arena_set_inferencer(agz_arena*, int, int, agz_net*) NYI("x")
int agz_arena_reset(agz_arena*, const uint8_t*) NYI("x")
int agz_arena_play(agz_arena*, int, int) NYI("x")
int agz_arena_begin_move(agz_arena*) NYI("x")
int agz_arena_simulate(agz_arena*, int) NYI("x")
int agz_arena_end_move(agz_arena*, int) NYI("x")
int agz_arena_get_stats(agz_arena*, agz_arena_stats*) NYI("x")
int agz_arena_get_game(agz_arena*, int, int32_t*, agz_game_state*) NYI("x")
int agz_arena_get_history(agz_arena*, int, int32_t*, int, int*) NYI("x")
int agz_arena_root_children(agz_arena*, int, int, int32_t*, uint32_t*, float*, float*, int, int*) NYI("x")
int agz_arena_tree_nodes(agz_arena*, int, int, int*) NYI("x")
int agz_arena_get_examples(agz_arena*, float*, float*, float*, int32_t*, int, int*) NYI("x")
int agz_arena_clear_examples(agz_arena*) NYI("x")
int agz_arena_examples_dev(agz_arena*, float**, float**, float**, int*) NYI("x")
}
