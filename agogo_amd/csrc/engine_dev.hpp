// Device side of the batched self-play engine: game rules, PUCT select, expand, backup, best move.
//
// One 64-lane wavefront (one workgroup) per game.  Every game owns two search trees (agent A, agent B —
// arena.go:42-58) stored as flat SoA node pools in HBM; the children of a node are created together
// (mcts/search.go:314-330) and therefore live in ONE contiguous block, so PUCT selection reads
// prior/visits/blackScores coalesced, 64 children per wave instruction.
//
// Bit-exactness contract (tests/test_engine_gpu.py): visit counts, blackScores and chosen moves equal the
// sequential CPU oracle.  All float arithmetic below mirrors the reference's operation order
// (mcts/node.go:147-237, mcts/search.go:259-339) with correctly-rounded +,-,*,/, a correctly rounded sqrt of OUR OWN (sqrt_cr, engine.hip: the toolchain's is not) and NO fma
// contraction (this translation unit is built with -ffp-contract=off).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/agz.h"

namespace agz {

constexpr int CELLS_PAD = 384;   // >= 361, multiple of 64
constexpr int RING = 8;          // boards kept for WQEncoder (encoding_helper.go:29-68)
constexpr int MAXPATH = 368;     // >= M*N + 2
constexpr int WAVE = 64;

enum LeafKind : int32_t { LEAF_NONE = 0, LEAF_EXPAND = 1, LEAF_TERMINAL = 2, LEAF_NULL = 3 };

struct GameCfg {
  int kind, m, n, k;
  int cells;        // m*n
  int A;            // game.ActionSpace(): moves without pass
  int max_moves;
  int encoder, F;
  float komi;
  int flip_in_tree;  // c4 (strict reference): Apply never flips nextToMove (game/c4/game.go:56-73)
  int pass_legal;    // wq, c4
  int go_like;       // komi, wq
  int has_passes;    // wq: Passes() counts; others return -1 / 0
};
struct MctsCfg {
  float PUCT;
  int maxDepth;  // M*N (tree.go:101)
  int RandomCount, Budget;
  uint32_t RandomMinVisits;
  float RandomTemperature;
  int DumbPass;
  float ResignPercentage;
  int PassPreference;
};

// All device buffers of an arena.  T = 2*G trees: tree(agent, g) = agent*G + g.
struct Dev {
  int G, T, cap;          // games, trees, node-pool capacity per (tree, pool)
  int V;                  // lanes per tree and round (agz_arena_set_parallel; 1 = the sequential search).  Per-simulation
                          // scratch below is indexed q = g*V + lane.
  // ---- game.State per game
  int8_t* board;          // [G][CELLS_PAD]
  int8_t* ring;           // [G][RING][CELLS_PAD]  board after move j at slot j % RING
  int32_t* to_move;       // nextToMove
  int32_t* ply;           // moves applied so far (len(history))
  int32_t* passes;        // wq Passes()
  int32_t* pass_count;    // Arena's passCount (arena.go:99-103)
  int32_t* ended;
  int32_t* winner;
  int32_t* a_is_black;
  int32_t* last_move;
  float* cap_b;           // komi: bs ; wq: captures by black
  float* cap_w;
  uint32_t* zhash;        // running zobrist hash (komi, wq)
  int16_t* moves;         // [G][max_moves + 4]   the GAME's history: moves applied to the board (LastMove/UndoLastMove/Fwd)
  int16_t* amoves;        // [G][max_moves + 4]   the ARENA's move list: every `best`, including a Resign and a Pass the game
                          //                      ignored (mnk/komi Apply of a pass is a no-op) — arena.go:125, agz_arena_get_history
  int32_t* n_amoves;      // [G]
  int32_t* hist_from;     // [G] earliest ply whose move is known in `moves` (0 for games played here; agz_arena_set_state may know fewer)
  const int32_t* ztable;  // [2*cells] zobrist keys
  // ---- trees
  float* prior;           // [T][2][cap]   P(s,a)  (Node.score)
  uint32_t* visits;       // N(s,a)
  float* bsum;            // blackScores
  int32_t* kids_off;      // first child index in the same pool, -1 = not expanded
  int16_t* kids_n;
  int16_t* nmove;         // game.Single of the node
  uint8_t* vl;            // virtualLoss flag (3.0 when set, node.go:248-260); only written when V > 1
  int32_t* n_nodes;       // [T]
  int32_t* cur_pool;      // [T]
  int32_t* has_root;      // [T]
  int32_t* has_prev;      // [T]  t.prev != nil
  int32_t* prev_ply;      // [T]
  int8_t* prev_board;     // [T][CELLS_PAD]
  int32_t* stalled;       // [T] a null simulation repeats forever in a deterministic search
  int32_t* overflow;      // [T]
  uint32_t* pc_hash;      // [T][max_moves+4] cachedPolicies keys (tree.go:75): (hash, move) per Search
  int16_t* pc_move;
  int32_t* pc_n;          // [T]
  uint64_t* rng;          // [T] SplitMix64 state of mcts.MCTS.rand (randomizeChildren)
  uint64_t* rng_game;     // [G] SplitMix64 state of Arena.r (colour draws on restart)
  // ---- per-simulation scratch
  int32_t* slot_of_game;  // [G] NN batch slot
  int32_t* prep_slot;     // [G] prepareRoot's compact batch: rank of game g among the roots the network has to evaluate (k_prep_compact)
  int32_t* leaf_kind;     // [G*V]
  int32_t* leaf_player;   // [G*V]
  int32_t* leaf_ply;      // [G*V]
  float* leaf_result;     // [G*V] terminal score; after k_expand: the value backed up by this lane
  int8_t* leaf_board;     // [G*V][CELLS_PAD]
  uint8_t* leaf_legal;    // [G*V][CELLS_PAD]  (index A = pass)
  int32_t* path;          // [G*V][MAXPATH]
  int32_t* path_len;      // [G*V]
  // lane rounds (V > 1): the expansion list of every lane, prepared in parallel (k_expand_prep) and committed in lane order
  float* exp_score;       // [G*V][CELLS_PAD] renormalised priors in child order (sorted)
  int16_t* exp_move;      // [G*V][CELLS_PAD]
  int32_t* exp_n;         // [G*V]
  float* exp_value;       // [G*V] the evaluation (Black's view) the expansion backs up
  // ---- counters [8]: sims_total, sims_nonnull, nn_evals, moves_played, games_finished, examples, tree_full
  unsigned long long* counters;
  // ---- examples
  float* ex_planes;       // [ex_cap][F*cells]
  float* ex_policy;       // [ex_cap][A+1]
  float* ex_value;        // [ex_cap]
  int32_t* ex_game;       // [ex_cap]
  int32_t* ex_prev;       // [ex_cap] previous example of the same game (-1 terminates)
  int32_t* ex_last;       // [G]
  uint8_t* ex_labelled;   // [ex_cap] 1 once the game of this example has ended and Value holds +1/-1/0 (arena.go:146-155)
  int32_t* ex_count;      // [1]
  int32_t* best_out;      // [G] result of a search-only end of move (agz_mcts_search: Search does not Apply, search.go:151-163)
  int ex_cap;
  int moves_stride;       // max_moves + 4
  // ---- AGZ_INF_CALLBACK: the leaf states a host inferencer is handed between k_select and k_expand
  float* cb_planes;       // [G*V][F*cells] the encoder's NCHW tensor of leaf q (what Agent.Infer encodes, agent.go:60-74); nullptr: no callback agent
  int cb_mask;            // bit a set: agent a holds a callback inferencer
};

enum { CNT_SIMS = 0, CNT_NONNULL = 1, CNT_EVALS = 2, CNT_MOVES = 3, CNT_GAMES = 4, CNT_EXAMPLES = 5, CNT_FULL = 6, CNT_A_WINS = 7, CNT_B_WINS = 8,
       CNT_DRAWS = 9, CNT_ILLEGAL = 10, CNT_DROPPED = 11, CNT_PATH = 12, CNT_KIDS = 13, CNT_PREP_EXPAND = 14, CNT_PATHMAX = 15, CNT_N = 16 };

// per-agent inferencer description passed to the expand kernel
struct InfDesc {
  int kind[2];            // AGZ_INF_*
  const float* policy[2]; // NN outputs [slots][policy_len]
  const float* value[2];
  int policy_len[2];
  int dummy_player[2];
};

__device__ __forceinline__ int opp(int p) { return p == AGZ_BLACK ? AGZ_WHITE : AGZ_BLACK; }

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
  return x;
}

// ---------------------------------------------------------------------------------------------------
// Shared-memory working set of one game
struct Sh {
  int8_t board[CELLS_PAD];
  int8_t ring[RING][CELLS_PAD];
  int32_t label[CELLS_PAD];   // connected-component label (min cell index) over equal-valued cells
  int32_t libs[CELLS_PAD];    // stone groups: number of distinct liberties (indexed by label)
  int32_t gsize[CELLS_PAD];   // component size (indexed by label)
  int32_t ghash[CELLS_PAD];   // stone groups: xor of zobrist keys
  int32_t touch[CELLS_PAD];   // empty components: bit0 touches black, bit1 touches white
  alignas(16) float fscore[CELLS_PAD];    // expansion: scores (read four at a time)
  int32_t fmove[CELLS_PAD];
  int32_t misc[16];
};

__device__ __forceinline__ int nbr(const GameCfg& c, int i, int d) {
  // adjacents {0,1},{1,0},{0,-1},{-1,0} in (X=row, Y=col): komi/game.go:404-409, wq.go:336-341
  int x = i / c.n, y = i - x * c.n;
  switch (d) {
    case 0: return (y + 1 < c.n) ? i + 1 : -1;
    case 1: return (x + 1 < c.m) ? i + c.n : -1;
    case 2: return (y - 1 >= 0) ? i - 1 : -1;
    default: return (x - 1 >= 0) ? i - c.n : -1;
  }
}

// Label all cells by connected component of equal value (stones AND empties), count liberties of stone
// groups, sizes, group zobrist xor, and which colours each empty region touches.
// Replaces the per-move flood fills of wq.go:237-290 / komi/game.go:348-402 (nolib) with one wave-parallel
// min-label propagation; results (capture sets, legal sets) are identical — see tests.
// empties = false: only STONE groups are labelled (empty cells keep their own index).  Captures and legality need nothing else
// (liberties are counted from the empty cells themselves), and label propagation then takes as many rounds as the widest stone
// group is across instead of the widest empty region (tens of rounds on an open 19x19 board); area scoring needs empties = true.
__device__ void analyse(const GameCfg& c, Sh& s, const int32_t* ztable, int lane, bool empties = true) {
  __syncthreads();
  for (int i = lane; i < c.cells; i += WAVE) {
    s.label[i] = i; s.libs[i] = 0; s.gsize[i] = 0; s.ghash[i] = 0; s.touch[i] = 0;
  }
  __syncthreads();
  for (int iter = 0; iter < 1024; iter++) {
    int changed = 0;
    for (int i = lane; i < c.cells; i += WAVE) {
      int v = s.board[i];
      if (v == AGZ_NONE && !empties) continue;
      int l0 = s.label[i];
      int l = l0;
#pragma unroll
      for (int d = 0; d < 4; d++) {
        int a = nbr(c, i, d);
        if (a >= 0 && s.board[a] == v) { int la = s.label[a]; l = la < l ? la : l; }
      }
      int ll = s.label[l];  // pointer jump (cell l belongs to the same component)
      l = ll < l ? ll : l;
      if (l < l0) { atomicMin(&s.label[i], l); changed = 1; }
    }
    if (!__syncthreads_or(changed)) break;
  }
  // at the fixpoint every cell of a component carries the component's minimum cell index
  for (int i = lane; i < c.cells; i += WAVE) {
    int v = s.board[i];
    int L = s.label[i];
    atomicAdd(&s.gsize[L], 1);
    if (v == AGZ_NONE) {
      int seen[4];
      int ns = 0;
#pragma unroll
      for (int d = 0; d < 4; d++) {
        int a = nbr(c, i, d);
        if (a < 0) continue;
        int va = s.board[a];
        if (va == AGZ_NONE) continue;
        atomicOr(&s.touch[L], va == AGZ_BLACK ? 1 : 2);
        int la = s.label[a];
        bool dup = false;
        for (int q = 0; q < ns; q++) dup |= (seen[q] == la);
        if (!dup) { seen[ns] = la; ns++; }
      }
      for (int q = 0; q < ns; q++) atomicAdd(&s.libs[seen[q]], 1);
    } else if (ztable) {
      atomicXor(&s.ghash[L], ztable[2 * i + (v == AGZ_BLACK ? 0 : 1)]);
    }
  }
  __syncthreads();
}

__device__ __forceinline__ bool has_empty_nbr(const GameCfg& c, const Sh& s, int i) {
  bool r = false;
#pragma unroll
  for (int d = 0; d < 4; d++) { int a = nbr(c, i, d); r |= (a >= 0 && s.board[a] == AGZ_NONE); }
  return r;
}
// does a stone of `player` at empty cell i capture something?  (check(): komi/game.go:316-338, wq.go:205-223)
__device__ __forceinline__ bool captures_any(const GameCfg& c, const Sh& s, int i, int player) {
  int o = opp(player);
  bool r = false;
#pragma unroll
  for (int d = 0; d < 4; d++) {
    int a = nbr(c, i, d);
    r |= (a >= 0 && s.board[a] == o && s.libs[s.label[a]] == 1);
  }
  return r;
}
// Check() for a board move under the reference rule set: empty and (captures or has an empty neighbour) —
// the suicide test runs BEFORE the stone is placed (komi/game.go:340-344), so friendly liberties do not count.
__device__ __forceinline__ bool go_legal(const GameCfg& c, const Sh& s, int i, int player) {
  return s.board[i] == AGZ_NONE && (captures_any(c, s, i, player) || has_empty_nbr(c, s, i));
}

// Apply a board move of a go-like game on the analysed board.  Returns `taken` with the reference's
// duplicate counting (one dead group touched on two sides is listed twice: SURVEY App. C c4b); *hash is
// updated with one xor per LISTED stone (so duplicates cancel, as in the reference).
__device__ int go_apply(const GameCfg& c, Sh& s, const int32_t* ztable, int cell, int player, uint32_t* hash, int lane) {
  int o = opp(player);
  int taken = 0;
  uint32_t h = *hash;
  int cap[4] = {-1, -1, -1, -1};
#pragma unroll
  for (int d = 0; d < 4; d++) {
    int a = nbr(c, cell, d);
    if (a >= 0 && s.board[a] == o) {
      int L = s.label[a];
      if (s.libs[L] == 1) { cap[d] = L; taken += s.gsize[L]; h ^= (uint32_t)s.ghash[L]; }
    }
  }
  __syncthreads();
  for (int i = lane; i < c.cells; i += WAVE) {
    if (s.board[i] == o) {
      int L = s.label[i];
      if (L == cap[0] || L == cap[1] || L == cap[2] || L == cap[3]) s.board[i] = AGZ_NONE;
    }
  }
  if (lane == 0) s.board[cell] = (int8_t)player;
  if (ztable) h ^= (uint32_t)ztable[2 * cell + (player == AGZ_BLACK ? 0 : 1)];
  *hash = h;
  __syncthreads();
  return taken;
}

// Tromp-Taylor area score of both colours from the analysed board (completion of wq Game.Score)
__device__ void area_scores(const GameCfg& c, Sh& s, int lane, float* black, float* white) {
  int b = 0, w = 0;
  for (int i = lane; i < c.cells; i += WAVE) {
    int v = s.board[i];
    if (v == AGZ_BLACK) b++;
    else if (v == AGZ_WHITE) w++;
    else {
      int t = s.touch[s.label[i]];
      if (t == 1) b++; else if (t == 2) w++;
    }
  }
  for (int o = 32; o > 0; o >>= 1) { b += __shfl_xor(b, o, 64); w += __shfl_xor(w, o, 64); }
  *black = (float)b; *white = (float)w;
}

// ---- mnk / c4 rules (serial, lane-uniform; boards are tiny) -----------------------------------------
// game/mnk/mnk.go:226-295 with its quirks
__device__ bool mnk_is_winner(const GameCfg& c, const int8_t* board, int colour) {
  int m = c.m, n = c.n, k = c.k;
  for (int i = 0; i < m; i++) {
    int rc = 0;
    for (int j = 0; j < n; j++) { if (board[i * n + j] == colour) rc++; else rc--; }
    if (rc >= k) return true;
  }
  for (int j = 0; j < n; j++) {
    int cnt = 0;
    for (int i = 0; i * n + j < m * n; i++) { if (board[i * n + j] == colour) cnt++; else cnt = 0; }
    if (cnt >= k) return true;
  }
  for (int i = 0; i < m; i++)
    for (int j = 0; n - j > n - k && j < n; j++) {
      int idx = i * n + j, dc = 0;
      while (board[idx] == colour) { dc++; if (dc >= k) return true; idx += n + 1; if (idx >= m * n) break; }
    }
  for (int i = 0; i < m; i++)
    for (int j = n - 1; j >= k - 1; j--) {
      int idx = i * n + j, dc = 0;
      while (board[idx] == colour) { dc++; if (dc >= k) return true; idx += n - 1; if (idx >= m * n) break; }
    }
  return false;
}
// game/c4/c4.go:82-192
__device__ int c4_check_win(const GameCfg& c, const int8_t* b) {
  int rows = c.m, cols = c.n, nw = c.k;
  const int dxs[4] = {0, 1, -1, 1}, dys[4] = {1, 0, 1, 1};  // vertical, horizontal, TLBR, TRBL
  for (int dir = 0; dir < 4; dir++)
    for (int x = 0; x < cols; x++)
      for (int y = 0; y < rows; y++) {
        int col = b[y * cols + x];
        if (col == AGZ_NONE) continue;
        bool w = true;
        for (int i = 0; i < nw; i++) {
          int xx = x + dxs[dir] * i, yy = y + dys[dir] * i;
          if (xx >= 0 && xx < cols && yy < rows) { if (b[yy * cols + xx] != col) w = false; } else w = false;
        }
        if (w) return col;
      }
  return AGZ_NONE;
}
__device__ __forceinline__ int c4_drop_row(const GameCfg& c, const int8_t* b, int col) {  // c4.go:59-70
  for (int r = c.m - 1; r >= 0; r--) if (b[r * c.n + col] == AGZ_NONE) return r;
  return -1;
}

// FNV-1a over "None"/"Black"/"White" (game/mnk/mnk.go:76-82, game/c4/game.go:203-210)
__device__ uint32_t fnv_board(const GameCfg& c, const int8_t* board) {
  uint32_t h = 2166136261u;
  for (int i = 0; i < c.cells; i++) {
    int v = board[i];
    const char* sname = v == AGZ_BLACK ? "Black" : (v == AGZ_WHITE ? "White" : "None");
    for (; *sname; ++sname) { h ^= (uint8_t)*sname; h *= 16777619u; }
  }
  return h;
}

}  // namespace agz
