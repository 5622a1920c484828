// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points (ctypes) over the CPU restatement.
// Loaded only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <memory>

#include "arena.hpp"
#include "train.hpp"
#include "examples.hpp"
#include "learn.hpp"

using namespace oracle;

namespace {
StatePtr make_game(int kind, int m, int n, int k, double komi) {
  switch (kind) {
    case 0: return std::make_shared<MNK>(m, n, k);
    case 1: return std::make_shared<C4>(m, n, k);
    case 2: return std::make_shared<Komi>(m, n, k);
    case 3: return std::make_shared<WQ>(m, 0, komi);
  }
  return nullptr;
}
GameEncoder make_enc(int enc) {
  if (enc == 1) return [](const State& s) { return WQEncoder(s); };
  return [](const State& s) { return EncodeTwoPlane(s); };
}
std::vector<Colour>& raw_board(State* s) {
  if (auto* g = dynamic_cast<MNK*>(s)) return g->board;
  if (auto* g = dynamic_cast<C4*>(s)) return g->data;
  if (auto* g = dynamic_cast<Komi*>(s)) return g->board;
  return dynamic_cast<WQ*>(s)->board.data;
}
struct ArenaBox {
  std::unique_ptr<Arena> arena;
  std::unique_ptr<Inferencer> inf[2];
  int kind, m, n, k, enc;
  double komi;
};
}  // namespace

extern "C" {

int orc_round(int a) { return dual_round(a); }

// ---- games ----
void* orc_game_new(int kind, int m, int n, int k, double komi) {
  StatePtr g = make_game(kind, m, n, k, komi);
  if (!g) return nullptr;
  return new StatePtr(g);
}
void orc_game_free(void* h) { delete (StatePtr*)h; }
void orc_game_set_board(void* h, const int32_t* b, int n) {
  auto& v = raw_board(((StatePtr*)h)->get());
  for (int i = 0; i < n && i < (int)v.size(); i++) v[i] = b[i];
}
void orc_game_get_board(void* h, int32_t* b) {
  const auto& v = (*(StatePtr*)h)->Board();
  for (size_t i = 0; i < v.size(); i++) b[i] = v[i];
}
void orc_game_set_to_move(void* h, int p) { (*(StatePtr*)h)->SetToMove(p); }
int orc_game_to_move(void* h) { return (*(StatePtr*)h)->ToMove(); }
int orc_game_move_number(void* h) { return (*(StatePtr*)h)->MoveNumber(); }
int orc_game_passes(void* h) { return (*(StatePtr*)h)->Passes(); }
int orc_game_action_space(void* h) { return (*(StatePtr*)h)->ActionSpace(); }
uint32_t orc_game_hash(void* h) { return (*(StatePtr*)h)->Hash(); }
int orc_game_check(void* h, int player, int move) { return (*(StatePtr*)h)->Check(PlayerMove{player, move}) ? 1 : 0; }
// game.State.Apply; the handle follows the returned state
void orc_game_apply(void* h, int player, int move) {
  StatePtr* sp = (StatePtr*)h;
  *sp = (*sp)->Apply(PlayerMove{player, move});
}
// komi: Game.Apply then (taken, err) as komi_test.go:186-190 reads them. returns taken, or -1 on error
int orc_komi_apply(void* h, int player, int move) {
  Komi* g = dynamic_cast<Komi*>(((StatePtr*)h)->get());
  if (!g) return -2;
  g->Apply(PlayerMove{player, move});
  return g->err ? -1 : g->taken;
}
// wq: Board.Apply as wq_test.go:206 calls it. returns taken, or -1 on error
int orc_wq_board_apply(void* h, int player, int move) {
  WQ* g = dynamic_cast<WQ*>(((StatePtr*)h)->get());
  if (!g) return -2;
  return g->board.Apply(PlayerMove{player, move});
}
float orc_wq_board_score(void* h, int player) {  // Board.Score (wq.go:173-202)
  WQ* g = dynamic_cast<WQ*>(((StatePtr*)h)->get());
  return g ? g->board.Score(player) : -1.f;
}
float orc_game_score(void* h, int player) { return (*(StatePtr*)h)->Score(player); }
int orc_game_ended(void* h, int* winner) {
  Player w = None;
  bool e = (*(StatePtr*)h)->Ended(&w);
  if (winner) *winner = w;
  return e ? 1 : 0;
}
int orc_mnk_is_winner(void* h, int player) {
  MNK* g = dynamic_cast<MNK*>(((StatePtr*)h)->get());
  return g && g->isWinner(player) ? 1 : 0;
}
void* orc_game_clone(void* h) { return new StatePtr((*(StatePtr*)h)->Clone()); }
int orc_game_eq(void* a, void* b) { return (*(StatePtr*)a)->Eq(((StatePtr*)b)->get()) ? 1 : 0; }
void orc_game_reset(void* h) { (*(StatePtr*)h)->Reset(); }
void orc_game_undo(void* h) { (*(StatePtr*)h)->UndoLastMove(); }
void orc_game_fwd(void* h) { (*(StatePtr*)h)->Fwd(); }
int orc_game_encode(void* h, int enc, float* out, int cap) {
  std::vector<float> r = make_enc(enc)(**(StatePtr*)h);
  if ((int)r.size() > cap) return -(int)r.size();
  memcpy(out, r.data(), r.size() * sizeof(float));
  return (int)r.size();
}

// ---- dual net ----
void* orc_net_new(int K, int L, int FC, int BatchSize, int W, int H, int F, int A, int bn_mode, float bn_eps) {
  DualConfig c;
  c.K = K; c.SharedLayers = L; c.FC = FC; c.BatchSize = BatchSize; c.Width = W; c.Height = H; c.Features = F;
  c.ActionSpace = A; c.bn_mode = bn_mode; c.bn_eps = bn_eps;
  if (!c.IsValid()) return nullptr;
  return new Dual(c);
}
void orc_net_free(void* h) { delete (Dual*)h; }
int orc_net_num_params(void* h) { return (int)((Dual*)h)->params.size(); }
int64_t orc_net_param_size(void* h, int i) { return (int64_t)((Dual*)h)->params.at(i).v.size(); }
const char* orc_net_param_name(void* h, int i) { return ((Dual*)h)->params.at(i).name.c_str(); }
void orc_net_get_param(void* h, int i, float* out) {
  auto& v = ((Dual*)h)->params.at(i).v;
  memcpy(out, v.data(), v.size() * sizeof(float));
}
void orc_net_set_param(void* h, int i, const float* in) {
  auto& v = ((Dual*)h)->params.at(i).v;
  memcpy(v.data(), in, v.size() * sizeof(float));
}
void orc_net_set_bn_stats(void* h, int bi, const float* mean, const float* var, int C) {
  auto& st = ((Dual*)h)->bn.at(bi);
  st.mean.assign(mean, mean + C);
  st.var.assign(var, var + C);
}
void orc_net_init_random(void* h, uint64_t seed) { ((Dual*)h)->InitRandom(seed); }
double orc_net_flops_per_eval(void* h) { return ((Dual*)h)->FlopsPerEval(); }
// planes [B,F,H,W] -> policy [B,A], value [B]; each board evaluated independently with the row-0 parameters
void orc_net_infer(void* h, const float* planes, int B, float* policy, float* value) {
  Dual* d = (Dual*)h;
  size_t per = (size_t)d->conf.Features * d->HW();
  for (int b = 0; b < B; b++) {
    std::vector<float> p;
    float v;
    d->Infer(planes + per * b, &p, &v);
    memcpy(policy + (size_t)b * d->conf.ActionSpace, p.data(), p.size() * sizeof(float));
    value[b] = v;
  }
}

// ---- arena (one game, two agents) ----
void* orc_arena_new(int kind, int m, int n, int k, double komi, int enc, float PUCT, int M, int N, int RandomCount,
                    int Budget, uint32_t RandomMinVisits, float RandomTemperature, int DumbPass,
                    float ResignPercentage, int PassPreference, uint64_t seed, int max_moves) {
  StatePtr g = make_game(kind, m, n, k, komi);
  if (!g) return nullptr;
  MCTSConfig c;
  c.PUCT = PUCT; c.M = M; c.N = N; c.RandomCount = RandomCount; c.Budget = Budget; c.RandomMinVisits = RandomMinVisits;
  c.RandomTemperature = RandomTemperature; c.DumbPass = DumbPass != 0; c.ResignPercentage = ResignPercentage;
  c.PassPreference = PassPreference;
  if (!c.IsValid()) return nullptr;
  auto* box = new ArenaBox();
  box->kind = kind; box->m = m; box->n = n; box->k = k; box->komi = komi; box->enc = enc;
  if (max_moves <= 0) max_moves = 2 * m * n;
  box->arena.reset(new Arena(g, c, make_enc(enc), seed, max_moves));
  return box;
}
void orc_arena_free(void* h) { delete (ArenaBox*)h; }
// kind: AGZ_INF_* ; net = orc_net handle for AGZ_INF_NET; dummy_player for AGZ_INF_DUMMY; policy_len for HASH/UNIFORM
int orc_arena_set_inferencer(void* h, int agent, int kind, void* net, int dummy_player, int policy_len) {
  ArenaBox* b = (ArenaBox*)h;
  Inferencer* inf = nullptr;
  int A = b->arena->game->ActionSpace();
  switch (kind) {
    case 0: if (!net) return -1; inf = new NetInferencer((Dual*)net, make_enc(b->enc)); break;
    case 1: inf = new DummyInferer(A, dummy_player); break;
    case 2: inf = new ScriptNN(); break;
    case 3: inf = new HashNN(policy_len > 0 ? policy_len : A + 1); break;
    case 4: inf = new UniformNN(policy_len > 0 ? policy_len : 25); break;
    default: return -1;
  }
  b->inf[agent].reset(inf);
  (agent == 0 ? b->arena->A : b->arena->B).nn = inf;
  return 0;
}
int orc_arena_set_callback(void* h, int agent, infer_cb cb, void* user, int policy_len) {
  ArenaBox* b = (ArenaBox*)h;
  Inferencer* inf = new CallbackInferencer(cb, user, make_enc(b->enc), policy_len);
  b->inf[agent].reset(inf);
  (agent == 0 ? b->arena->A : b->arena->B).nn = inf;
  return 0;
}
// MCTSConfig::Parallel (lanes per tree and round); call before orc_arena_begin
void orc_arena_set_parallel(void* h, int lanes) { ((ArenaBox*)h)->arena->conf.Parallel = lanes < 1 ? 1 : lanes; }
void orc_arena_begin(void* h, int a_is_black) { ((ArenaBox*)h)->arena->Begin(a_is_black); }
int orc_arena_step(void* h, int record) { return ((ArenaBox*)h)->arena->Step(record != 0) ? 1 : 0; }
// apply an externally chosen move for the player to move; returns 1 if the game continues, 0 if it ended, -1 if illegal
int orc_arena_apply_move(void* h, int move) {
  Arena* a = ((ArenaBox*)h)->arena.get();
  if (a->ended) return 0;
  size_t before = a->moves.size();
  Single m = (Single)move;
  bool cont = a->Step(false, &m);
  if (a->moves.size() == before) return -1;
  return cont ? 1 : 0;
}
// plays up to n_moves plies (<=0: to the end); returns plies played
int orc_arena_play(void* h, int n_moves, int record) {
  Arena* a = ((ArenaBox*)h)->arena.get();
  int played = 0;
  while (!a->ended && (n_moves <= 0 || played < n_moves)) { a->Step(record != 0); played++; }
  return played;
}
int orc_arena_history(void* h, int32_t* moves, int cap) {
  Arena* a = ((ArenaBox*)h)->arena.get();
  int n = (int)a->moves.size();
  for (int i = 0; i < n && i < cap; i++) moves[i] = a->moves[i];
  return n;
}
// out: [to_move, move_number, passes, ended, winner, a_is_black]
void orc_arena_state(void* h, int32_t* board, int32_t* out) {
  Arena* a = ((ArenaBox*)h)->arena.get();
  if (board) { const auto& v = a->game->Board(); for (size_t i = 0; i < v.size(); i++) board[i] = v[i]; }
  out[0] = a->game->ToMove(); out[1] = a->game->MoveNumber(); out[2] = a->game->Passes();
  out[3] = a->ended ? 1 : 0; out[4] = a->winner; out[5] = a->A.player == Black ? 1 : 0;
}
// root children of an agent's tree in their current order (bestMove sorts them in place, search.go:353)
int orc_arena_root_children(void* h, int agent, int32_t* moves, uint32_t* visits, float* bscores, float* priors, int cap) {
  Arena* a = ((ArenaBox*)h)->arena.get();
  MCTS* t = (agent == 0 ? a->A : a->B).mcts.get();
  if (!t || t->root == nilNode) return 0;
  const auto& kids = t->children.at(t->root);
  int n = (int)kids.size();
  for (int i = 0; i < n && i < cap; i++) {
    const Node& nd = t->N(kids[i]);
    moves[i] = nd.move; visits[i] = nd.visits; bscores[i] = nd.blackScores; priors[i] = nd.score;
  }
  return n;
}
// out: [nnEvals, playouts, lastIter, nodes, rootVisits] ; root blackScores via *root_bs
void orc_arena_tree_stats(void* h, int agent, int64_t* out, float* root_bs) {
  Arena* a = ((ArenaBox*)h)->arena.get();
  MCTS* t = (agent == 0 ? a->A : a->B).mcts.get();
  out[0] = t->nnEvals; out[1] = t->playouts; out[2] = t->lastIter; out[3] = t->Nodes();
  out[4] = t->root == nilNode ? 0 : t->N(t->root).visits;
  if (root_bs) *root_bs = t->root == nilNode ? 0.f : t->N(t->root).blackScores;
}
int orc_arena_num_examples(void* h) { return (int)((ArenaBox*)h)->arena->examples.size(); }
void orc_arena_get_example(void* h, int i, float* board, float* policy, float* value) {
  const Example& e = ((ArenaBox*)h)->arena->examples.at(i);
  memcpy(board, e.Board.data(), e.Board.size() * sizeof(float));
  memcpy(policy, e.Policy.data(), e.Policy.size() * sizeof(float));
  *value = e.Value;
}
int orc_arena_example_sizes(void* h, int* board_len, int* policy_len) {
  Arena* a = ((ArenaBox*)h)->arena.get();
  if (a->examples.empty()) return 0;
  *board_len = (int)a->examples[0].Board.size();
  *policy_len = (int)a->examples[0].Policy.size();
  return 1;
}


// ---- mcts/example_test.go:74-103 pattern: ONE tree searched alternately for both players ----
struct ExampleBox {
  StatePtr g;
  std::unique_ptr<Inferencer> inf;
  std::unique_ptr<MCTS> t;
  Player player;
};
void* orc_example_new(int kind, int m, int n, int k, double komi, float PUCT, int Budget, int inf_kind, int policy_len,
                      int first_player, uint64_t seed) {
  auto* b = new ExampleBox();
  b->g = make_game(kind, m, n, k, komi);
  MCTSConfig c;
  c.PUCT = PUCT; c.M = m; c.N = n; c.Budget = Budget; c.DumbPass = true; c.PassPreference = DontPreferPass; c.RandomCount = 0;
  int A = b->g->ActionSpace();
  switch (inf_kind) {
    case 2: b->inf.reset(new ScriptNN()); break;
    case 3: b->inf.reset(new HashNN(policy_len > 0 ? policy_len : A + 1)); break;
    case 4: b->inf.reset(new UniformNN(policy_len > 0 ? policy_len : 25)); break;
    default: b->inf.reset(new DummyInferer(A, 0)); break;
  }
  b->t.reset(new MCTS(b->g, c, b->inf.get(), seed));
  b->player = first_player;
  return b;
}
void orc_example_free(void* h) { delete (ExampleBox*)h; }
// one loop iteration of Example(): returns best move; *ended / *winner evaluated after the Apply
int orc_example_turn(void* h, int* ended, int* winner) {
  ExampleBox* b = (ExampleBox*)h;
  Single best = b->t->Search(b->player);
  b->g = b->g->Apply(PlayerMove{b->player, best});
  b->t->SetGame(b->g);
  b->player = Opponent(b->player);
  Player w = None;
  *ended = b->g->Ended(&w) ? 1 : 0;
  *winner = w;
  return best;
}
int orc_example_root_children(void* h, int32_t* moves, uint32_t* visits, float* bscores, int cap) {
  MCTS* t = ((ExampleBox*)h)->t.get();
  const auto& kids = t->children.at(t->root);
  int n = (int)kids.size();
  for (int i = 0; i < n && i < cap; i++) { const Node& nd = t->N(kids[i]); moves[i] = nd.move; visits[i] = nd.visits; bscores[i] = nd.blackScores; }
  return n;
}
int64_t orc_example_nn_evals(void* h) { return ((ExampleBox*)h)->t->nnEvals; }


// ---- one mcts.MCTS on a caller-owned game.State: mcts.New / SetGame / Search / Policies / Nodes (tree.go:80-142, search.go:92) ----
struct MctsBox {
  std::unique_ptr<Inferencer> inf;
  std::unique_ptr<MCTS> t;
  int enc;
};
void* orc_mcts_new(void* game, int enc, float PUCT, int M, int N, int RandomCount, int Budget, uint32_t RandomMinVisits,
                   float RandomTemperature, int DumbPass, float ResignPercentage, int PassPreference, int lanes, int inf_kind,
                   int policy_len, uint64_t seed) {
  StatePtr g = *(StatePtr*)game;
  MCTSConfig c;
  c.PUCT = PUCT; c.M = M; c.N = N; c.RandomCount = RandomCount; c.Budget = Budget; c.RandomMinVisits = RandomMinVisits;
  c.RandomTemperature = RandomTemperature; c.DumbPass = DumbPass != 0; c.ResignPercentage = ResignPercentage;
  c.PassPreference = PassPreference; c.Parallel = lanes < 1 ? 1 : lanes;
  if (!c.IsValid()) return nullptr;
  auto* b = new MctsBox();
  b->enc = enc;
  int A = g->ActionSpace();
  switch (inf_kind) {
    case 2: b->inf.reset(new ScriptNN()); break;
    case 3: b->inf.reset(new HashNN(policy_len > 0 ? policy_len : A + 1)); break;
    case 4: b->inf.reset(new UniformNN(policy_len > 0 ? policy_len : 25)); break;
    default: b->inf.reset(new DummyInferer(A, 0)); break;
  }
  b->t.reset(new MCTS(g, c, b->inf.get(), seed));
  return b;
}
void orc_mcts_free(void* h) { delete (MctsBox*)h; }
void orc_mcts_set_callback(void* h, infer_cb cb, void* user, int policy_len) {
  MctsBox* b = (MctsBox*)h;
  b->inf.reset(new CallbackInferencer(cb, user, make_enc(b->enc), policy_len));
  b->t->nn = b->inf.get();
}
// SetGame (tree.go:120-124): the tree searches a CLONE of the caller's state (Agent.Search hands over its own game object,
// agent.go:77-80; the device copies the state too), so the caller's handle stays untouched by Search's SetToMove
void orc_mcts_set_game(void* h, void* game) { ((MctsBox*)h)->t->SetGame((*(StatePtr*)game)->Clone()); }
int orc_mcts_search(void* h, int player) { return ((MctsBox*)h)->t->Search(player); }
int orc_mcts_policies(void* h, void* game, float* out, int cap) {
  std::vector<float> p = ((MctsBox*)h)->t->Policies(**(StatePtr*)game);
  for (size_t i = 0; i < p.size() && (int)i < cap; i++) out[i] = p[i];
  return (int)p.size();
}
int orc_mcts_root_children(void* h, int32_t* moves, uint32_t* visits, float* bscores, float* priors, int cap) {
  MCTS* t = ((MctsBox*)h)->t.get();
  if (t->root == nilNode) return 0;
  const auto& kids = t->children.at(t->root);
  int n = (int)kids.size();
  for (int i = 0; i < n && i < cap; i++) {
    const Node& nd = t->N(kids[i]);
    moves[i] = nd.move; visits[i] = nd.visits; bscores[i] = nd.blackScores; priors[i] = nd.score;
  }
  return n;
}
// out: [nnEvals, playouts, lastIter, nodes]
void orc_mcts_stats(void* h, int64_t* out) {
  MCTS* t = ((MctsBox*)h)->t.get();
  out[0] = t->nnEvals; out[1] = t->playouts; out[2] = t->lastIter; out[3] = t->Nodes();
}

// Synthetic opening ("u uniformly-random legal moves from the empty board"): the arena's player to move plays
// the (z mod n_legal)-th legal board move in ascending cell order, z = SplitMix64 finaliser of (seed, game index, arena move
// count); no legal board move: Pass where the game has one, else nothing.  Restated by k_random_pick (engine.hip).
// returns the move played, or -32768 when none was
int orc_arena_random_move(void* h, uint64_t seed, int g) {
  Arena* a = ((ArenaBox*)h)->arena.get();
  if (a->ended) return -32768;
  Player pl = a->currentPlayer->player;
  int A = a->game->ActionSpace();
  std::vector<int> legal;
  for (int i = 0; i < A; i++) if (a->game->Check(PlayerMove{pl, (Single)i})) legal.push_back(i);
  Single mv;
  if (legal.empty()) {
    if (!a->game->Check(PlayerMove{pl, Pass})) return -32768;
    mv = Pass;
  } else {
    uint64_t z = seed + 0x9E3779B97F4A7C15ull * (uint64_t)(g + 1) + 0xD1B54A32D192ED03ull * (uint64_t)(a->moves.size() + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    mv = (Single)legal[(size_t)(z % (uint64_t)legal.size())];
  }
  a->Step(false, &mv);
  return mv;
}


// ---- dual.Train restatement (oracle/train.hpp) ----
static DualConfig mkconf(int K, int L, int FC, int BatchSize, int W, int H, int F, int A, float eps) {
  DualConfig c; c.K = K; c.SharedLayers = L; c.FC = FC; c.BatchSize = BatchSize; c.Width = W; c.Height = H; c.Features = F;
  c.ActionSpace = A; c.bn_mode = 0; c.bn_eps = eps; return c;
}
void* orc_train_new(int K, int L, int FC, int BatchSize, int W, int H, int F, int A, float eps) {
  DualConfig c = mkconf(K, L, FC, BatchSize, W, H, F, A, eps);
  if (!c.IsValid()) return nullptr;
  return new TrainNet<float>(c);
}
void orc_train_free(void* h) { delete (TrainNet<float>*)h; }
int orc_train_num_params(void* h) { return (int)((TrainNet<float>*)h)->P.size(); }
int64_t orc_train_param_size(void* h, int i) { return (int64_t)((TrainNet<float>*)h)->P.at(i).size(); }
const char* orc_train_param_name(void* h, int i) { return ((TrainNet<float>*)h)->names.at(i).c_str(); }
void orc_train_init_random(void* h, uint64_t seed) { ((TrainNet<float>*)h)->InitRandom(seed); }
void orc_train_get_param(void* h, int i, float* out) { auto& v = ((TrainNet<float>*)h)->P.at(i); memcpy(out, v.data(), v.size() * 4); }
void orc_train_set_param(void* h, int i, const float* in) { auto& v = ((TrainNet<float>*)h)->P.at(i); memcpy(v.data(), in, v.size() * 4); }
void orc_train_get_grad(void* h, int i, float* out) { auto& v = ((TrainNet<float>*)h)->G.at(i); memcpy(out, v.data(), v.size() * 4); }
// one batch: forward (training-mode BN) + backward; lr > 0 also applies the vanilla SGD step. returns the cost
float orc_train_batch(void* h, const float* planes, const float* Pi, const float* V, float lr) {
  TrainNet<float>* t = (TrainNet<float>*)h;
  float c = t->ForwardBackward(planes, Pi, V, true);
  if (lr > 0) t->Step(lr);
  return c;
}
// analytic gradients vs central finite differences, both in double. returns the max relative error over n_checks
// randomly chosen parameter entries (spread over all parameter tensors).
double orc_train_gradcheck(int K, int L, int FC, int BatchSize, int W, int H, int F, int A, uint64_t seed, int n_checks) {
  DualConfig c = mkconf(K, L, FC, BatchSize, W, H, F, A, 1e-5f);
  TrainNet<double> t(c);
  t.InitRandom(seed);
  SplitMix64 r(seed ^ 0x1234);
  for (auto& p : t.P) for (double& x : p) x *= 3.0;  // move away from the tiny Glorot scale so ReLUs are mixed
  for (size_t i = 0; i < t.P.size(); i++) if (t.kinds[i] == 1) for (double& x : t.P[i]) x = 0.5 + r.float64();
  size_t nx = (size_t)BatchSize * F * H * W;
  std::vector<double> X(nx), Pi((size_t)BatchSize * A), V(BatchSize);
  for (double& x : X) x = r.float64() * 2 - 1;
  for (int b = 0; b < BatchSize; b++) { Pi[(size_t)b * A + (r.next() % A)] = 1.0; V[b] = (double)((int)(r.next() % 3) - 1); }
  t.ForwardBackward(X.data(), Pi.data(), V.data(), true);
  auto G = t.G;
  double worst = 0;
  for (int k = 0; k < n_checks; k++) {
    size_t pi = (size_t)(k % t.P.size());
    size_t idx = (size_t)(r.next() % t.P[pi].size());
    double keep = t.P[pi][idx], eps = 1e-5;
    t.P[pi][idx] = keep + eps; double cp = t.ForwardBackward(X.data(), Pi.data(), V.data(), false);
    t.P[pi][idx] = keep - eps; double cm = t.ForwardBackward(X.data(), Pi.data(), V.data(), false);
    t.P[pi][idx] = keep;
    double num = (cp - cm) / (2 * eps), ana = G[pi][idx];
    double err = std::fabs(num - ana) / std::max(1e-7, std::fabs(num) + std::fabs(ana));
    if (std::fabs(num) + std::fabs(ana) > 1e-9 && err > worst) worst = err;
  }
  return worst;
}

// ---- example plumbing (oracle/examples.hpp) ----
int orc_rotate_board(const float* board, int m, int n, float* out) { return RotateBoard(board, m, n, out) ? 0 : -1; }
void* orc_exset_new(int F, int m, int n, int A1) { return new ExampleSet{F, m, n, A1, {}, {}, {}}; }
void orc_exset_free(void* h) { delete (ExampleSet*)h; }
void orc_exset_push(void* h, const float* boards, const float* policies, const float* values, int count) {
  ExampleSet* s = (ExampleSet*)h;
  for (int i = 0; i < count; i++) s->push(boards + (size_t)i * s->F * s->m * s->n, policies + (size_t)i * s->A1, values[i]);
}
int orc_exset_size(void* h) { return (int)((ExampleSet*)h)->size(); }
int orc_exset_augment_rotate(void* h) { return ((ExampleSet*)h)->AugmentRotate() ? 0 : -1; }
void orc_exset_get(void* h, float* boards, float* policies, float* values) {
  ExampleSet* s = (ExampleSet*)h;
  size_t xs = (size_t)s->F * s->m * s->n;
  for (size_t i = 0; i < s->size(); i++) {
    memcpy(boards + i * xs, s->board[i].data(), xs * 4);
    memcpy(policies + i * s->A1, s->policy[i].data(), (size_t)s->A1 * 4);
    values[i] = s->value[i];
  }
}
// prepareExamples; outputs must hold (size/BatchSize)*BatchSize rows (after the maxExamples cut). returns batches
int orc_exset_prepare(void* h, int BatchSize, int maxExamples, uint64_t seed, float* Xs, float* Pi, float* V) {
  ExampleSet* s = (ExampleSet*)h;
  std::vector<float> x, p, v;
  int batches = s->Prepare(BatchSize, maxExamples, seed, &x, &p, &v);
  memcpy(Xs, x.data(), x.size() * 4); memcpy(Pi, p.data(), p.size() * 4); memcpy(V, v.data(), v.size() * 4);
  return batches;
}


// ---- AZ.Learn restatement (oracle/learn.hpp).  out: per epoch 12 floats {epoch, examples, batches, cost, a_wins, a_loss, a_draw,
// b_wins, b_loss, b_draw, killedA, a_id}; returns the epochs completed (fewer than iters: "batches is nil"), -1 on a bad config.
int orc_learn_run(int kind, int m, int n, int k, double komi, int enc, int K, int L, int FC, int BatchSize, int F, int A,
                  float PUCT, int Budget, int RandomCount, float threshold, int maxExamples, int augment,
                  int sp_inf0, int sp_inf1, int ev_inf0, int ev_inf1, uint64_t seed,
                  int iters, int episodes, int nniters, int arenaGames, float* out) {
  LearnConfig c;
  c.kind = kind; c.m = m; c.n = n; c.k = k; c.komi = komi; c.enc = enc;
  c.nn = mkconf(K, L, FC, BatchSize, n, m, F, A, 1e-5f);
  c.mc.PUCT = PUCT; c.mc.M = m; c.mc.N = n; c.mc.Budget = Budget; c.mc.RandomCount = RandomCount;
  c.UpdateThreshold = threshold; c.MaxExamples = maxExamples; c.AugmentRotate = augment != 0;
  c.sp_inf[0] = sp_inf0; c.sp_inf[1] = sp_inf1; c.eval_inf[0] = ev_inf0; c.eval_inf[1] = ev_inf1; c.seed = seed;
  if (!c.nn.IsValid() || !c.mc.IsValid() || !learn_make_game(kind, m, n, k, komi)) return -1;
  Learner lr(c);
  lr.Learn(iters, episodes, nniters, arenaGames);
  for (size_t e = 0; e < lr.log.size(); e++) {
    const EpochLog& g = lr.log[e];
    float* o = out + 12 * e;
    o[0] = (float)g.epoch; o[1] = (float)g.examples; o[2] = (float)g.batches; o[3] = g.cost; o[4] = g.a_wins; o[5] = g.a_loss; o[6] = g.a_draw;
    o[7] = g.b_wins; o[8] = g.b_loss; o[9] = g.b_draw; o[10] = g.killedA ? 1.f : 0.f; o[11] = (float)g.a_id;
  }
  return (int)lr.log.size();
}
}  // extern "C"
