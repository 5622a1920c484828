// Batched self-play engine: n_games x { agogo.Arena + 2 x mcts.MCTS + game.State } resident in HBM.
//
// Reference path replaced:
//   Arena.Play loop           arena.go:96-138        -> agz_arena_play / begin_move / simulate / end_move
//   MCTS.Search               mcts/search.go:92-164  -> k_begin_move (updateRoot) + prepare + Budget x step + k_end_move
//   searchState.pipeline      mcts/search.go:209-257 -> k_select (descent+Apply) ; NN ; k_expand (expand+backup)
//   Node.Select / Evaluate    mcts/node.go:147-237   -> select_child()
//   expandAndSimulate         mcts/search.go:259-339 -> k_expand
//   bestMove / fancySort      mcts/search.go:341-390, mcts/utils.go:18-47 -> k_end_move
//   Policies / cachedPolicies mcts/tree.go:128-142   -> policy-cache list per tree
//   game.State (mnk,c4,komi,wq) game/*                -> engine_dev.hpp
//   encoders                  encoding_helper.go:29-68, cmd/tictactoe/main.go:26-47 -> encode_leaf()
//
// One simulation step = every unfinished game runs ONE pipeline() in its current agent's tree; the leaves of
// all games are coalesced into one batched network evaluation (the reference evaluates one leaf per call in
// row 0 of an ActionSpace-row batch, dualnet/meta.go:175-189).  Searches are sequential per tree, so the
// statistics equal the sequential oracle bit for bit; parallelism comes from thousands of trees.
//
// Bound: these kernels are HBM/latency-bound integer + scalar-float work (select reads 12 B per child,
// backup writes 8 B per path node); they are sized to stay off the critical path of the conv tower.
#include <algorithm>
#include <chrono>
#include <climits>
#include <cstring>
#include <vector>

#include "engine_dev.hpp"
#include "net.hpp"

namespace agz {

__device__ __forceinline__ size_t pool_base(const Dev& d, int t, int pool) { return ((size_t)t * 2 + pool) * d.cap; }

// Evaluate(): mcts/node.go:147-159 (virtual loss is never observable in a sequential search)
__device__ __forceinline__ float evaluate(float bs, uint32_t visits, int player) {
  float score = __fdiv_rn(bs, (float)visits);
  if (player == AGZ_WHITE) score = __fsub_rn(1.0f, score);
  return score;
}

// Correctly rounded float32 square root.  On this toolchain (ROCm 7.2, __clang_hip_math.h) __fsqrt_rn IS __ocml_native_sqrt_f32 and sqrtf
// compiles to the same 1-ulp instruction sequence whatever -fhip-fp32-correctly-rounded-divide-sqrt says: 2 526 173 of the integers 1 .. 2^24
// come out one ulp off (scripts/probes/select_arith_probe.hip; sqrt(300.f) = 0x418a9066 instead of 0x418a9067) — and rounds 1-5 computed PUCT's
// sqrt(parentVisits) with it.  One ulp in the numerator flips an argmax only at a near-tie: no test saw it until round 6's narrow-tree fuzz
// (value 0, equal priors: 0.5 / 154 against (1/14) / 22 at 300 parent visits) did.  Here: the native estimate, bracketed with EXACT double
// products (a 24-bit significand squared has 48 bits) so that r <= sqrt(x) < next(r), then x against the exact square of the midpoint
// (25 bits -> 50): the nearer neighbour, for every x.  (float division IS correctly rounded on the device: the same probe, 4 M pairs.)
__device__ __forceinline__ float sqrt_cr(float x) {
  if (!(x > 0.f)) return x == 0.f ? x : __builtin_amdgcn_sqrtf(x);   // +-0 as they are, negatives / NaN -> NaN (visits sums never are)
  float r = __builtin_amdgcn_sqrtf(x);
  while ((double)r * (double)r > (double)x) r = __uint_as_float(__float_as_uint(r) - 1u);
  float rn = __uint_as_float(__float_as_uint(r) + 1u);
  while ((double)rn * (double)rn <= (double)x) { r = rn; rn = __uint_as_float(__float_as_uint(r) + 1u); }
  const double mid = 0.5 * ((double)r + (double)rn);
  return (double)x > mid * mid ? rn : r;
}

// Node.Select: mcts/node.go:170-237.  64 lanes over the contiguous child block; strict '>' keeps the first maximum.
__device__ int select_child(const Dev& d, size_t base, int off, int n, int player, float PUCT, int lane, bool use_vl, int* move_out = nullptr) {
  // The whole child block goes to registers in ONE batch of loads (n <= CELLS_PAD: six children per lane): visits, sums, priors,
  // virtual-loss marks and the children's moves — the parent-visit sum and the argmax then run without touching memory again, and the
  // caller gets the chosen child's move with it (the descent was four dependent memory round trips per level; now two).
  constexpr int PER = CELLS_PAD / WAVE;
  uint32_t cv[PER]; float cb[PER], cp[PER]; int cm[PER]; uint8_t cl[PER];
  const size_t cb0 = base + off;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int i = lane + k * WAVE;
    const bool ok = i < n;
    const size_t o = cb0 + (ok ? i : 0);
    cv[k] = d.visits[o]; cb[k] = d.bsum[o]; cp[k] = d.prior[o]; cm[k] = d.nmove[o];
    cl[k] = use_vl ? d.vl[o] : 0;
    if (!ok) cv[k] = 0;
  }
  uint32_t pv = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) pv += cv[k];
  for (int o = 32; o > 0; o >>= 1) pv += __shfl_xor(pv, o, 64);
  float numerator = sqrt_cr((float)pv);   // math32.Sqrt (node.go:209): correctly rounded — NOT __fsqrt_rn / sqrtf, see sqrt_cr above
  float best = -INFINITY;
  int idx = -1, mv = 0;
#pragma unroll
  for (int k = 0; k < PER; k++) {
    const int i = lane + k * WAVE;
    if (i < n) {
      uint32_t v = cv[k];
      float bs = cb[k];
      // Evaluate (node.go:147-159): only White's view includes the stored virtual loss (3.0 while a lane of this round is below)
      if (use_vl && player == AGZ_WHITE) bs = __fadd_rn(bs, cl[k] ? 3.0f : 0.0f);
      float psa = cp[k];
      float qsa = evaluate(bs, v, player);
      float denominator = __fadd_rn(1.0f, (float)v);
      float lastTerm = __fdiv_rn(numerator, denominator);
      float puct = __fmul_rn(__fmul_rn(PUCT, psa), lastTerm);
      float usa = __fadd_rn(qsa, puct);
      if (usa > best) { best = usa; idx = i; mv = cm[k]; }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    float ob = __shfl_xor(best, o, 64);
    int oi = __shfl_xor(idx, o, 64);
    int om = __shfl_xor(mv, o, 64);
    bool take = (ob > best) || (ob == best && oi >= 0 && (idx < 0 || oi < idx));
    if (take) { best = ob; idx = oi; mv = om; }
  }
  if (move_out) *move_out = mv;
  return idx;  // -1: the reference panics "Cannot return nil" (node.go:232-234)
}

// sort.Sort(byScore) as a STABLE descending sort of s.fscore[0..n): rank = #greater + #equal-before, for the (up
// to CELLS_PAD / 64) entries lane + 64 k of this lane.  For non-negative, non-NaN scores (every prior that comes out of a softmax / the
// renormalisation) the float order is the order of the bit patterns, so "(sj > si) || (sj == si && j < i)" is ONE unsigned 64-bit
// comparison of the keys (bits << 32 | ~index); a lane counts for all its entries at once, independent chains per broadcast LDS read.
// (One wave ranking ~300 moves with dependent float-compare chains took 36 of k_expand's 47 us at batch 1.)  Anything else — a NaN or
// a negative score — takes the float comparisons as written.  Call with all 64 lanes; entries past n get rank 0 (unused).
constexpr int RANK_PER = CELLS_PAD / WAVE;
__device__ void stable_desc_ranks(const float* fscore, int n, int lane, int* rnk) {
  bool odd = false;
  for (int i2 = lane; i2 < n; i2 += WAVE) { const unsigned bts = __float_as_uint(fscore[i2]); odd |= (bts & 0x80000000u) != 0u || (bts & 0x7fffffffu) > 0x7f800000u; }
  if (!__syncthreads_or(odd ? 1 : 0)) {
    unsigned long long key[RANK_PER];
#pragma unroll
    for (int k = 0; k < RANK_PER; k++) {
      const int i2 = lane + k * WAVE;
      key[k] = i2 < n ? (((unsigned long long)__float_as_uint(fscore[i2]) << 32) | (unsigned)(0x7fffffff - i2)) : ~0ull;
      rnk[k] = 0;
    }
    int jj = 0;
    for (; jj + 4 <= n; jj += 4) {
      const float4 v4 = *reinterpret_cast<const float4*>(&fscore[jj]);
      const unsigned long long k0 = ((unsigned long long)__float_as_uint(v4.x) << 32) | (unsigned)(0x7fffffff - jj);
      const unsigned long long k1 = ((unsigned long long)__float_as_uint(v4.y) << 32) | (unsigned)(0x7fffffff - jj - 1);
      const unsigned long long k2 = ((unsigned long long)__float_as_uint(v4.z) << 32) | (unsigned)(0x7fffffff - jj - 2);
      const unsigned long long k3 = ((unsigned long long)__float_as_uint(v4.w) << 32) | (unsigned)(0x7fffffff - jj - 3);
#pragma unroll
      for (int k = 0; k < RANK_PER; k++) rnk[k] += (int)(k0 > key[k]) + (int)(k1 > key[k]) + (int)(k2 > key[k]) + (int)(k3 > key[k]);
    }
    for (; jj < n; jj++) {
      const unsigned long long kj = ((unsigned long long)__float_as_uint(fscore[jj]) << 32) | (unsigned)(0x7fffffff - jj);
#pragma unroll
      for (int k = 0; k < RANK_PER; k++) rnk[k] += (int)(kj > key[k]);
    }
  } else {
#pragma unroll
    for (int k = 0; k < RANK_PER; k++) {
      const int i = lane + k * WAVE;
      int rank = 0;
      if (i < n) {
        const float si = fscore[i];
        for (int j = 0; j < n; j++) { const float sj = fscore[j]; rank += (sj > si) || (sj == si && j < i); }
      }
      rnk[k] = rank;
    }
  }
}

// ---- state in LDS -----------------------------------------------------------------------------------
struct St {
  int to_move, ply, passes;
  uint32_t hash;
};

// game.State.Apply for a move that is known to be legal; updates s.board / ring / st.  Returns captures taken.
__device__ int apply_move(const GameCfg& c, const Dev& d, Sh& s, St& st, int mv, bool use_ring, int lane) {
  int player = st.to_move;
  int taken = 0;
  if (c.kind == AGZ_GAME_MNK) {  // mnk.go:122-142
    if (lane == 0) s.board[mv] = (int8_t)player;
  } else if (c.kind == AGZ_GAME_C4) {  // c4/game.go:56-73, c4.go:47-70
    if (mv != AGZ_PASS) {
      int r = c4_drop_row(c, s.board, mv);
      __syncthreads();
      if (lane == 0 && r >= 0) s.board[r * c.n + mv] = (int8_t)player;
    }
  } else {  // komi/game.go:104-130,277-313 ; wq/game.go:81-92 (+pass completion)
    if (mv != AGZ_PASS) {
      // A capture needs an opponent group whose ONLY liberty is mv.  If every opponent stone next to mv has another empty neighbour
      // of its own, no group can be in that state and Apply is "place the stone" — no component labelling of the whole board (the
      // common case inside the tree; the batch-1 descent spent most of its time labelling).  Otherwise: the full analysis, as before.
      const int o = opp(player);
      bool maybe = false;
#pragma unroll
      for (int dd = 0; dd < 4; dd++) {
        const int a = nbr(c, mv, dd);
        if (a >= 0 && s.board[a] == o) {
          bool other = false;
#pragma unroll
          for (int e = 0; e < 4; e++) { const int b2 = nbr(c, a, e); other |= (b2 >= 0 && b2 != mv && s.board[b2] == AGZ_NONE); }
          maybe |= !other;
        }
      }
      if (maybe) {
        analyse(c, s, d.ztable, lane, false);
        taken = go_apply(c, s, d.ztable, mv, player, &st.hash, lane);
      } else {
        __syncthreads();                       // every lane has read the board
        if (lane == 0) s.board[mv] = (int8_t)player;
        if (d.ztable) st.hash ^= (uint32_t)d.ztable[2 * mv + (player == AGZ_BLACK ? 0 : 1)];
      }
      st.passes = 0;
    } else {
      st.passes++;
    }
  }
  __syncthreads();
  if (c.flip_in_tree) st.to_move = opp(player);
  st.ply++;
  if (use_ring) {
    int slot = st.ply % RING;
    for (int i = lane; i < c.cells; i += WAVE) s.ring[slot][i] = s.board[i];
    __syncthreads();
  }
  return taken;
}

__device__ __forceinline__ int move_number(const GameCfg& c, int ply) { return c.kind == AGZ_GAME_C4 ? 1 : ply; }  // c4/game.go:52

// legality of every action for `player` on the board in s (go-like boards must be analysed)
__device__ void legal_mask(const GameCfg& c, const Sh& s, int player, uint8_t* out, int lane) {
  for (int i = lane; i < c.A; i += WAVE) {
    bool ok;
    if (c.kind == AGZ_GAME_MNK) ok = s.board[i] == AGZ_NONE;                         // mnk.go:102-120
    else if (c.kind == AGZ_GAME_C4) ok = s.board[i] == AGZ_NONE;                     // top cell of column i empty (c4.go:59-70)
    else ok = go_legal(c, s, i, player);                                               // komi/game.go:78-102
    out[i] = ok ? 1 : 0;
  }
  if (lane == 0) out[c.A] = c.pass_legal ? 1 : 0;
}

// Encoders.  Output is the network's input tile for this leaf: padded NHWC [Hp][Wp][32].
__device__ void encode_nhwc(const GameCfg& c, const Sh& s, const St& st, float* out, int lane) {
  const int Wp = c.n + 2;
  for (int i = lane; i < c.cells; i += WAVE) {
    int h = i / c.n, w = i - h * c.n;
    float* o = out + ((size_t)(h + 1) * Wp + (w + 1)) * 32;
    if (c.encoder == AGZ_ENC_TWOPLANE) {  // cmd/tictactoe/main.go:26-47
      int v = s.board[i];
      o[0] = v == AGZ_BLACK ? 1.f : (v == AGZ_WHITE ? -1.f : 0.001f);
      o[1] = st.to_move == AGZ_BLACK ? 1.f : (st.to_move == AGZ_WHITE ? -1.f : 0.f);
    } else {  // WQEncoder, encoding_helper.go:29-68
      int mn = move_number(c, st.ply);
      bool nb = st.to_move == AGZ_BLACK;
      float vb[8], vw[8];                       // the mover's planes 0..7 and the opponent's
#pragma unroll
      for (int k = 1; k < 8; k++) {
        int j = mn - k;  // board after move j (Historical(h), h = mn-1-k, needs h > 0)
        float e = 0.f;
        if (j >= 2) { int v = s.ring[j % RING][i]; e = v == AGZ_BLACK ? 1.f : (v == AGZ_WHITE ? -1.f : 0.f); }
        vb[k - 1] = e;
        vw[k - 1] = j >= 2 ? -e : 0.f;  // vecf32.Scale(retVal, -1): empty cells become -0.0
      }
      vb[7] = 0.f; vw[7] = 0.f;
      // channels 0..7 = black planes, 8..15 = white planes when Black is to move (swapped otherwise), 16 / 17 the colour planes,
      // 18..19 padding (zero like the rest of the 32-channel slot): five 16-byte stores per cell instead of 18 scalar ones
      float4* o4 = reinterpret_cast<float4*>(o);
      const float* lo8 = nb ? vb : vw;
      const float* hi8 = nb ? vw : vb;
      o4[0] = make_float4(lo8[0], lo8[1], lo8[2], lo8[3]);
      o4[1] = make_float4(lo8[4], lo8[5], lo8[6], lo8[7]);
      o4[2] = make_float4(hi8[0], hi8[1], hi8[2], hi8[3]);
      o4[3] = make_float4(hi8[4], hi8[5], hi8[6], hi8[7]);
      o4[4] = make_float4(nb ? 1.f : 0.f, nb ? 0.f : -1.f, 0.f, 0.f);
    }
  }
}
// same encoders in the reference's flat NCHW [F][cells] layout (training examples, arena.go:106)
__device__ void encode_nchw(const GameCfg& c, const Sh& s, const St& st, float* out, int lane) {
  for (int i = lane; i < c.cells; i += WAVE) {
    if (c.encoder == AGZ_ENC_TWOPLANE) {
      int v = s.board[i];
      out[i] = v == AGZ_BLACK ? 1.f : (v == AGZ_WHITE ? -1.f : 0.001f);
      out[c.cells + i] = st.to_move == AGZ_BLACK ? 1.f : (st.to_move == AGZ_WHITE ? -1.f : 0.f);
    } else {
      int mn = move_number(c, st.ply);
      bool nb = st.to_move == AGZ_BLACK;
      int blackStart = nb ? 0 : 8, whiteStart = nb ? 8 : 0;
      for (int k = 1; k < 8; k++) {
        int j = mn - k;
        float e = 0.f;
        if (j >= 2) { int v = s.ring[j % RING][i]; e = v == AGZ_BLACK ? 1.f : (v == AGZ_WHITE ? -1.f : 0.f); }
        out[(blackStart + k - 1) * c.cells + i] = e;
        out[(whiteStart + k - 1) * c.cells + i] = j >= 2 ? -e : 0.f;  // -0.0 for empty cells, as vecf32.Scale gives
      }
      out[7 * c.cells + i] = 0.f; out[15 * c.cells + i] = 0.f;
      out[16 * c.cells + i] = nb ? 1.f : 0.f;
      out[17 * c.cells + i] = nb ? 0.f : -1.f;
    }
  }
}

__device__ void load_state(const GameCfg& c, const Dev& d, int g, Sh& s, St& st, bool use_ring, int lane) {
  // whole padded rows in 16-byte words (the pad cells are zero on both sides): 4 loads per lane instead of 54 byte loads
  {
    const uint4* gb = reinterpret_cast<const uint4*>(d.board + (size_t)g * CELLS_PAD);
    uint4* sb = reinterpret_cast<uint4*>(s.board);
    for (int i = lane; i < CELLS_PAD / 16; i += WAVE) sb[i] = gb[i];
    if (use_ring) {
      const uint4* gr = reinterpret_cast<const uint4*>(d.ring + (size_t)g * RING * CELLS_PAD);
      uint4* sr = reinterpret_cast<uint4*>(&s.ring[0][0]);
      for (int i = lane; i < RING * CELLS_PAD / 16; i += WAVE) sr[i] = gr[i];
    }
  }
  st.to_move = d.to_move[g];
  st.ply = d.ply[g];
  st.passes = d.passes[g];
  st.hash = d.zhash[g];
  __syncthreads();
}

__device__ __forceinline__ int agent_of(const Dev& d, int g) {
  // the Arena's currentPlayer (arena.go:81-89,226-233): A moves when its colour is to move
  int tm = d.to_move[g];
  bool a_black = d.a_is_black[g] != 0;
  return ((tm == AGZ_BLACK) == a_black) ? 0 : 1;
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_reset(Dev d, GameCfg c, const uint8_t* a_is_black, unsigned long long seed,
                                               unsigned long long tree_seed) {
  int g = blockIdx.x, lane = threadIdx.x;
  for (int i = lane; i < CELLS_PAD; i += WAVE) d.board[(size_t)g * CELLS_PAD + i] = 0;
  for (int i = lane; i < RING * CELLS_PAD; i += WAVE) d.ring[(size_t)g * RING * CELLS_PAD + i] = 0;
  if (lane == 0) {
    int ab;
    if (a_is_black) ab = a_is_black[g] ? 1 : 0;
    else {  // a.r.Intn(2) == 0 -> A is Black (arena.go:81); the build's SplitMix64 stream
      unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(g + 1);
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
      ab = (z >> 63) == 0 ? 1 : 0;
    }
    d.a_is_black[g] = ab;
    d.to_move[g] = AGZ_BLACK;  // the agent holding Black starts: game.SetToMove(currentPlayer.Player), arena.go:91
    d.ply[g] = 0; d.passes[g] = 0; d.pass_count[g] = 0; d.ended[g] = 0; d.winner[g] = AGZ_NONE; d.n_amoves[g] = 0;
    d.last_move[g] = AGZ_PASS; d.cap_b[g] = 0.f; d.cap_w[g] = 0.f; d.zhash[g] = 0; d.ex_last[g] = -1; d.hist_from[g] = 0;
    for (int l = 0; l < d.V; l++) d.leaf_kind[(size_t)g * d.V + l] = LEAF_NONE;
    d.rng_game[g] = seed * 0x9E3779B97F4A7C15ull + (unsigned long long)g * 0xD1B54A32D192ED03ull + 7ull;
    for (int a = 0; a < 2; a++) {
      int t = a * d.G + g;
      d.n_nodes[t] = 0; d.cur_pool[t] = 0; d.has_root[t] = 0; d.has_prev[t] = 0; d.prev_ply[t] = 0; d.stalled[t] = 0;
      d.overflow[t] = 0; d.pc_n[t] = 0;
      d.rng[t] = (tree_seed + (unsigned long long)g) * 2ull + 1ull + (unsigned long long)a;  // mcts.New's rand (tree.go:84): stream of the oracle's Arena(seed+g)
    }
  }
}

// updateRoot + newRootState: mcts/search.go:424-500.  Tree reuse re-roots by copying the kept subtree into
// the tree's other pool in BFS block order (the reference frees the siblings into a freelist, tree.go:183-209;
// node identities are not observable, statistics and child order are).
__global__ __launch_bounds__(64) void k_begin_move(Dev d, GameCfg c, MctsCfg mc) {
  __shared__ Sh s;
  int g = blockIdx.x, lane = threadIdx.x;
  if (d.ended[g]) return;
  int agent = agent_of(d, g);
  int t = agent * d.G + g;
  int player = d.to_move[g];
  St st;
  load_state(c, d, g, s, st, false, lane);
  // a search is judged by what IT adds: the overflow mark of an earlier search (the tree was re-rooted and compacted since) must not
  // hide this one's from CNT_FULL (a wall-clock search ends on it, a Budget search reports AGZ_E_TREE_FULL)
  if (lane == 0) { d.stalled[t] = 0; d.overflow[t] = 0; }
  int pool = d.cur_pool[t];
  size_t base = pool_base(d, t, pool);
  bool ok = d.has_root[t] && d.has_prev[t] && c.kind != AGZ_GAME_C4;  // c4: Eq() can never hold (App. C c2)
  int depth = 0, node = 0;
  if (ok) {
    int prev_ply = d.prev_ply[t];
    depth = st.ply - prev_ply;
    if (depth < 0 || prev_ply < d.hist_from[g]) ok = false;   // (UndoLastMove needs the moves since prev: a state set from outside may not carry them)
    if (ok && c.kind != AGZ_GAME_WQ) {
      // tmp := current.Clone(); UndoLastMove x depth (removes the stone only, mnk.go:184-189, komi/game.go:201-206);
      // tmp.Eq(prev) compares boards.
      int bad = 0;
      for (int i = lane; i < c.cells; i += WAVE) {
        int v = s.board[i];
        for (int k = 0; k < depth; k++) { int mv = d.moves[(size_t)g * d.moves_stride + st.ply - 1 - k]; if (mv == i) v = AGZ_NONE; }
        if (v != d.prev_board[(size_t)t * CELLS_PAD + i]) bad = 1;
      }
      if (__syncthreads_or(bad)) ok = false;
    }
    for (int k = 0; ok && k < depth; k++) {  // replay LastMove()s, findChild (node.go:285-296)
      int mv = d.moves[(size_t)g * d.moves_stride + prev_ply + k];
      int off = d.kids_off[base + node], n = off >= 0 ? d.kids_n[base + node] : 0;
      int found = 0x7fffffff;
      for (int i = lane; i < n; i += WAVE) if (d.nmove[base + off + i] == mv && i < found) found = i;
      for (int o = 32; o > 0; o >>= 1) { int f2 = __shfl_xor(found, o, 64); found = f2 < found ? f2 : found; }
      if (found == 0x7fffffff) ok = false; else node = off + found;
    }
  }
  if (ok) {
    if (node != 0) {
      // compaction copy of the subtree rooted at `node` into the other pool
      size_t nb = pool_base(d, t, pool ^ 1);
      if (lane == 0) {
        d.prior[nb] = d.prior[base + node]; d.visits[nb] = d.visits[base + node]; d.bsum[nb] = d.bsum[base + node];
        d.kids_off[nb] = d.kids_off[base + node]; d.kids_n[nb] = d.kids_n[base + node]; d.nmove[nb] = d.nmove[base + node];
      }
      __syncthreads();
      int new_n = 1, scan = 0;
      while (scan < new_n) {
        int hi = min(scan + WAVE, new_n);
        int i = scan + lane;
        int ooff = (i < hi) ? d.kids_off[nb + i] : -1;
        unsigned long long m = __ballot(ooff >= 0);
        while (m) {
          int src_lane = __ffsll((long long)m) - 1;
          m &= m - 1;
          int so = __shfl(ooff, src_lane, 64);
          int ni = scan + src_lane;
          int cnt = d.kids_n[nb + ni];
          for (int k = lane; k < cnt; k += WAVE) {
            d.prior[nb + new_n + k] = d.prior[base + so + k];
            d.visits[nb + new_n + k] = d.visits[base + so + k];
            d.bsum[nb + new_n + k] = d.bsum[base + so + k];
            d.kids_off[nb + new_n + k] = d.kids_off[base + so + k];  // still an OLD-pool offset until scanned
            d.kids_n[nb + new_n + k] = d.kids_n[base + so + k];
            d.nmove[nb + new_n + k] = d.nmove[base + so + k];
          }
          if (lane == 0) d.kids_off[nb + ni] = new_n;
          new_n += cnt;
        }
        __syncthreads();
        scan = hi;
      }
      if (lane == 0) { d.cur_pool[t] = pool ^ 1; d.n_nodes[t] = new_n; }
    }
  } else {
    // fresh root: Pass if legal, else the first legal move (search.go:476-487); pool restarts empty
    int rootmv = AGZ_PASS;
    if (!c.pass_legal) {
      if (c.go_like) analyse(c, s, nullptr, lane, false);
      int first = 0x7fffffff;
      for (int i = lane; i < c.A; i += WAVE) {
        bool legal = c.go_like ? go_legal(c, s, i, player) : (s.board[i] == AGZ_NONE);
        if (legal && i < first) first = i;
      }
      for (int o = 32; o > 0; o >>= 1) { int f2 = __shfl_xor(first, o, 64); first = f2 < first ? f2 : first; }
      rootmv = first == 0x7fffffff ? AGZ_PASS : first;
    }
    if (lane == 0) {
      d.prior[base] = 0.f; d.visits[base] = 1; d.bsum[base] = 0.f; d.kids_off[base] = -1; d.kids_n[base] = 0;
      d.nmove[base] = (int16_t)rootmv; d.n_nodes[t] = 1; d.has_root[t] = 1;
    }
  }
  if (lane == 0) d.has_prev[t] = 0;  // t.searchState.prev = nil (search.go:490)
}

// pipeline() descent: mcts/search.go:209-257 up to (and excluding) the network call.
// prep != 0 runs prepareRoot (search.go:392-408) instead: the root itself is the leaf if it is expandable.
__global__ __launch_bounds__(64) void k_select(Dev d, GameCfg c, MctsCfg mc, float* act_in0, float* act_in1, int prep, int nl) {
  __shared__ Sh s;
  int g = blockIdx.x, lane = threadIdx.x;
  const size_t q0 = (size_t)g * d.V;
  if (d.ended[g]) { if (lane < nl) d.leaf_kind[q0 + lane] = LEAF_NONE; return; }
  int agent = agent_of(d, g);
  int t = agent * d.G + g;
  if (d.stalled[t]) { if (lane < nl) { d.leaf_kind[q0 + lane] = LEAF_NULL; d.path_len[q0 + lane] = 0; } return; }
  const bool use_ring = c.encoder == AGZ_ENC_WQ;
  const bool use_vl = d.V > 1;   // lanes of one round see each other's stored virtual loss (oracle: MCTS::parallelRound)
  size_t base = pool_base(d, t, d.cur_pool[t]);
  for (int l = 0; l < nl; l++) {
    const size_t q = q0 + l;
    if (l > 0) { __threadfence(); __syncthreads(); }   // the previous lane's vl marks
    St st;
    load_state(c, d, g, s, st, use_ring, lane);
    int32_t* path = d.path + q * MAXPATH;
    int node = 0, depth = 1, plen = 1, kind = LEAF_NONE;
    int kids_seen = 0;
    float result = 0.f;
    if (lane == 0) path[0] = 0;
    while (true) {
      if (depth > mc.maxDepth) { kind = LEAF_NULL; plen--; break; }  // search.go:211-215: returns before addVirtualLoss
      if (use_vl && lane == 0) d.vl[base + node] = 1;              // n.addVirtualLoss() (search.go:222)
      int off = d.kids_off[base + node];
      if (off < 0) {  // IsExpandable(0)
        if (c.has_passes && st.passes >= 2) {
          if (prep) { kind = LEAF_TERMINAL; result = 0.f; }  // expandAndSimulate returns (0,false); root.Update(0)
          else {  // combinedScore, utils.go:62-67
            analyse(c, s, nullptr, lane);
            float b, w;
            area_scores(c, s, lane, &b, &w);
            result = __fsub_rn(__fsub_rn(b, w), c.komi);
            kind = LEAF_TERMINAL;
          }
        } else {
          kind = LEAF_EXPAND;
        }
        break;
      }
      if (prep) { kind = LEAF_NONE; break; }  // root already has children: prepareRoot does nothing
      int n = d.kids_n[base + node];
      kids_seen += n;
      int mv;
      int ci = select_child(d, base, off, n, st.to_move, mc.PUCT, lane, use_vl, &mv);
      if (ci < 0) { kind = LEAF_NULL; break; }
      int child = off + ci;
      apply_move(c, d, s, st, mv, use_ring, lane);  // children were created from legal moves of this very state
      if (lane == 0) path[plen] = child;
      plen++;
      node = child;
      depth++;
    }
    if (kind == LEAF_EXPAND) {
      if (c.go_like) analyse(c, s, nullptr, lane, false);
      legal_mask(c, s, st.to_move, d.leaf_legal + q * CELLS_PAD, lane);
      for (int i = lane; i < c.cells; i += WAVE) d.leaf_board[q * CELLS_PAD + i] = s.board[i];
      float* act = agent == 0 ? act_in0 : act_in1;
      // NN slots are lane-major ([lane][game slot]) so that a round of nl lanes is one dense batch of nl*G rows
      if (act) encode_nhwc(c, s, st, act + ((size_t)l * d.G + d.slot_of_game[g]) * (c.m + 2) * (c.n + 2) * 32, lane);
      if (d.cb_planes && ((d.cb_mask >> agent) & 1)) encode_nchw(c, s, st, d.cb_planes + q * (size_t)c.F * c.cells, lane);   // AGZ_INF_CALLBACK
    }
    if (lane == 0) {
      d.leaf_kind[q] = kind;
      d.leaf_player[q] = st.to_move;
      d.leaf_ply[q] = move_number(c, st.ply);
      d.leaf_result[q] = result;
      d.path_len[q] = plen;
      if (prep && kind == LEAF_EXPAND) atomicAdd(&d.counters[CNT_PREP_EXPAND], 1ull);   // roots prepareRoot has to evaluate
      if (!prep) {   // measurement only: nodes on the path and children read by Select, summed over simulations
        atomicAdd(&d.counters[CNT_PATH], (unsigned long long)plen);
        atomicAdd(&d.counters[CNT_KIDS], (unsigned long long)kids_seen);
        atomicMax(&d.counters[CNT_PATHMAX], (unsigned long long)plen);   // the longest descent so far (a step lasts as long as its longest path)
      }
    }
    __syncthreads();
  }
}

// ---- lane rounds (V > 1): the same round as k_select / k_expand with the lane-INDEPENDENT work moved out of the sequential
// lane loop.  What must run lane after lane is only what reads or writes shared tree state: the PUCT descent (it sees the
// virtual-loss marks of the earlier lanes), and — after the network — child-block allocation, backup and undoVirtualLoss.
// The descent needs no board: the mover alternates with the path (c4: never), Passes() follows from the moves on the path,
// the depth from its length.  Replaying the path on the board, the leaf's legal set, its input planes, and the expansion
// list (renormalise + stable sort) depend on that lane's path / network row alone: one workgroup per (game, lane).
// Results are bit-identical to the fused kernels (tests/test_parallel_lanes_gpu.py runs both against the oracle).
__global__ __launch_bounds__(64) void k_select_paths(Dev d, GameCfg c, MctsCfg mc, int prep, int nl) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const size_t q0 = (size_t)g * d.V;
  if (d.ended[g]) { if (lane < nl) d.leaf_kind[q0 + lane] = LEAF_NONE; return; }
  const int agent = agent_of(d, g);
  const int t = agent * d.G + g;
  if (d.stalled[t]) { if (lane < nl) { d.leaf_kind[q0 + lane] = LEAF_NULL; d.path_len[q0 + lane] = 0; } return; }
  const size_t base = pool_base(d, t, d.cur_pool[t]);
  const int to_move0 = d.to_move[g], ply0 = d.ply[g], passes0 = d.passes[g];
  for (int l = 0; l < nl; l++) {
    const size_t q = q0 + l;
    if (l > 0) { __threadfence(); __syncthreads(); }   // the previous lane's vl marks
    int32_t* path = d.path + q * MAXPATH;
    int node = 0, depth = 1, plen = 1, kind = LEAF_NONE, kids_seen = 0;
    int to_move = to_move0, ply = ply0, passes = passes0;
    if (lane == 0) path[0] = 0;
    while (true) {
      if (depth > mc.maxDepth) { kind = LEAF_NULL; plen--; break; }
      if (lane == 0) d.vl[base + node] = 1;
      const int off = d.kids_off[base + node];
      if (off < 0) { kind = (c.has_passes && passes >= 2) ? LEAF_TERMINAL : LEAF_EXPAND; break; }
      if (prep) { kind = LEAF_NONE; break; }
      const int n = d.kids_n[base + node];
      kids_seen += n;
      int mv;
      const int ci = select_child(d, base, off, n, to_move, mc.PUCT, lane, true, &mv);
      if (ci < 0) { kind = LEAF_NULL; break; }
      const int child = off + ci;
      // what Apply does to the scalars the descent reads (apply_move): mover, Passes(), move count
      if (c.go_like) { if (mv != AGZ_PASS) passes = 0; else passes++; }
      if (c.flip_in_tree) to_move = opp(to_move);
      ply++;
      if (lane == 0) path[plen] = child;
      plen++;
      node = child;
      depth++;
    }
    if (lane == 0) {
      d.leaf_kind[q] = kind;
      d.leaf_player[q] = to_move;
      d.leaf_ply[q] = move_number(c, ply);
      d.leaf_result[q] = 0.f;
      d.path_len[q] = plen;
      if (prep && kind == LEAF_EXPAND) atomicAdd(&d.counters[CNT_PREP_EXPAND], 1ull);
      if (!prep) {
        atomicAdd(&d.counters[CNT_PATH], (unsigned long long)plen);
        atomicAdd(&d.counters[CNT_KIDS], (unsigned long long)kids_seen);
        atomicMax(&d.counters[CNT_PATHMAX], (unsigned long long)plen);   // the longest descent so far (a step lasts as long as its longest path)
      }
    }
    __syncthreads();
  }
}

// one workgroup per (game, lane): replay the lane's path on the board, then the leaf work of k_select
__global__ __launch_bounds__(64) void k_leaf(Dev d, GameCfg c, MctsCfg mc, float* act_in0, float* act_in1, int prep, int nl) {
  __shared__ Sh s;
  const int g = blockIdx.x / nl, l = blockIdx.x - g * nl, lane = threadIdx.x;
  const size_t q = (size_t)g * d.V + l;
  const int kind = d.leaf_kind[q];
  if (kind != LEAF_EXPAND && kind != LEAF_TERMINAL) return;
  if (kind == LEAF_TERMINAL && prep) return;                 // root.Update(0): no board needed
  const int agent = agent_of(d, g);
  const int t = agent * d.G + g;
  const bool use_ring = c.encoder == AGZ_ENC_WQ;
  const size_t base = pool_base(d, t, d.cur_pool[t]);
  St st;
  load_state(c, d, g, s, st, use_ring, lane);
  const int32_t* path = d.path + q * MAXPATH;
  const int plen = d.path_len[q];
  for (int j = 1; j < plen; j++) apply_move(c, d, s, st, d.nmove[base + path[j]], use_ring, lane);
  if (kind == LEAF_TERMINAL) {   // combinedScore, utils.go:62-67
    analyse(c, s, nullptr, lane);
    float b, w;
    area_scores(c, s, lane, &b, &w);
    if (lane == 0) d.leaf_result[q] = __fsub_rn(__fsub_rn(b, w), c.komi);
    return;
  }
  if (c.go_like) analyse(c, s, nullptr, lane, false);
  legal_mask(c, s, st.to_move, d.leaf_legal + q * CELLS_PAD, lane);
  for (int i = lane; i < c.cells; i += WAVE) d.leaf_board[q * CELLS_PAD + i] = s.board[i];
  float* act = agent == 0 ? act_in0 : act_in1;
  if (act) encode_nhwc(c, s, st, act + ((size_t)l * d.G + d.slot_of_game[g]) * (c.m + 2) * (c.n + 2) * 32, lane);
  if (d.cb_planes && ((d.cb_mask >> agent) & 1)) encode_nchw(c, s, st, d.cb_planes + q * (size_t)c.F * c.cells, lane);
}

// one workgroup per (game, lane): the evaluation and the renormalised, sorted expansion list of an expandable leaf
__global__ __launch_bounds__(64) void k_expand_prep(Dev d, GameCfg c, MctsCfg mc, InfDesc inf, int nl) {
  __shared__ Sh s;
  const int g = blockIdx.x / nl, l = blockIdx.x - g * nl, lane = threadIdx.x;
  const size_t q = (size_t)g * d.V + l;
  if (d.ended[g] || d.leaf_kind[q] != LEAF_EXPAND) return;
  const int agent = agent_of(d, g);
  const int player = d.leaf_player[q];
  const uint8_t* legal = d.leaf_legal + q * CELLS_PAD;
  const int ik = inf.kind[agent];
  const int plen_pol = inf.policy_len[agent];
  const float* pol = nullptr;
  float value = 0.f;
  uint32_t ph = 0;
  if (ik == AGZ_INF_NET) {
    const int slot = l * d.G + d.slot_of_game[g];
    pol = inf.policy[agent] + (size_t)slot * plen_pol;
    value = inf.value[agent][slot];
  } else if (ik == AGZ_INF_DUMMY) {
    const int dp = inf.dummy_player[agent];
    value = dp == 1 ? 1.f : (dp == 2 ? -1.f : 0.f);
  } else if (ik == AGZ_INF_SCRIPT) {
    const int mn = d.leaf_ply[q];
    value = (mn == 0 || mn == 1 || mn == 5) ? 0.5f : 0.f;
  } else if (ik == AGZ_INF_HASH) {
    uint32_t hsh = 0;
    for (int i = lane; i < c.cells; i += WAVE) hsh += mix32((uint32_t)i * 4u + (uint32_t)d.leaf_board[q * CELLS_PAD + i] + 1u);
    for (int o = 32; o > 0; o >>= 1) hsh += __shfl_xor(hsh, o, 64);
    hsh += mix32(0xABCD0000u + (uint32_t)player);
    ph = hsh;
    value = (float)(mix32(hsh ^ 0xDEADBEEFu) >> 8) * (1.0f / 16777216.0f);
  } else {
    value = 1 / 25.0f;
  }
  auto policy_at = [&](int i) -> float {
    switch (ik) {
      case AGZ_INF_NET: return pol[i];
      case AGZ_INF_DUMMY: return __fdiv_rn(1.f, (float)plen_pol);
      case AGZ_INF_SCRIPT: {
        const int mn = d.leaf_ply[q];
        const int8_t cell[9] = {4, 0, 2, 6, 3, 5, 1, 7, 8};
        if (mn >= 0 && mn < 9 && i == cell[mn]) return (mn & 1) ? 0.1f : 0.9f;
        return 0.f;
      }
      case AGZ_INF_HASH: return (float)((mix32(ph + (uint32_t)i * 0x9E3779B9u) >> 8) + 1u) * (1.0f / 16777216.0f);
      default: return 1 / 25.0f;
    }
  };
  if (player == AGZ_WHITE) value = __fsub_rn(1.f, value);
  int n = 0;
  for (int b0 = 0; b0 < c.A; b0 += WAVE) {
    const int i = b0 + lane;
    const bool ok = i < c.A && legal[i];
    const unsigned long long m = __ballot(ok);
    if (ok) {
      const int pos = n + __popcll(m & ((1ull << lane) - 1ull));
      s.fscore[pos] = policy_at(i);
      s.fmove[pos] = i;
    }
    n += __popcll(m);
  }
  if (legal[c.A]) {
    if (lane == 0) { s.fscore[n] = policy_at(plen_pol - 1); s.fmove[n] = AGZ_PASS; }
    n++;
  }
  __syncthreads();
  float legalSum = 0.f;
  if (lane == 0) { for (int i = 0; i < n; i++) legalSum = __fadd_rn(legalSum, s.fscore[i]); }
  legalSum = __shfl(legalSum, 0, 64);
  if (legalSum > 1.401298464e-45f) {
    for (int i = lane; i < n; i += WAVE) s.fscore[i] = __fdiv_rn(s.fscore[i], legalSum);
  } else {
    const float prob = __fdiv_rn(1.f, (float)n);
    for (int i = lane; i < n; i += WAVE) s.fscore[i] = prob;
  }
  __syncthreads();
  {   // stable descending rank sort (as k_expand)
    int rnk[RANK_PER];
    stable_desc_ranks(s.fscore, n, lane, rnk);
#pragma unroll
    for (int k = 0; k < RANK_PER; k++) {
      const int i = lane + k * WAVE;
      if (i < n) { d.exp_score[q * CELLS_PAD + rnk[k]] = s.fscore[i]; d.exp_move[q * CELLS_PAD + rnk[k]] = (int16_t)s.fmove[i]; }
    }
  }
  if (lane == 0) { d.exp_n[q] = n; d.exp_value[q] = value; }
}

// lanes in order: child blocks, Update along the path, undoVirtualLoss (the sequential half of k_expand)
__global__ __launch_bounds__(64) void k_expand_commit(Dev d, GameCfg c, MctsCfg mc, int prep, int nl) {
  const int g = blockIdx.x, lane = threadIdx.x;
  if (d.ended[g]) return;
  const int agent = agent_of(d, g);
  const int t = agent * d.G + g;
  const size_t q0 = (size_t)g * d.V;
  const size_t base = pool_base(d, t, d.cur_pool[t]);
  int n_null = 0, n_lanes = 0;
  unsigned have_mask = 0;
  for (int l = 0; l < nl; l++) {
    const size_t q = q0 + l;
    if (l > 0) { __threadfence(); __syncthreads(); }
    const int kind = d.leaf_kind[q];
    if (kind == LEAF_NONE) continue;
    n_lanes++;
    if (!prep && lane == 0) atomicAdd(&d.counters[CNT_SIMS], 1ull);
    const int32_t* path = d.path + q * MAXPATH;
    const int plen = d.path_len[q];
    if (kind == LEAF_NULL) {
      n_null++;
      for (int j = lane; j < plen; j += WAVE) d.vl[base + path[j]] = 0;
      continue;
    }
    const int node = path[plen - 1];
    float result = d.leaf_result[q];
    bool have = true, follower = false;
    if (kind == LEAF_EXPAND && l > 0) {
      for (int l2 = 0; l2 < l && !follower; l2++) {
        const size_t q2 = q0 + l2;
        if (d.leaf_kind[q2] == LEAF_EXPAND && d.path_len[q2] == plen && d.path[q2 * MAXPATH + plen - 1] == node) {
          follower = true;
          result = d.leaf_result[q2];
          have = (have_mask >> l2) & 1u;
        }
      }
    }
    if (kind == LEAF_EXPAND && !follower) {
      const int n = d.exp_n[q];
      result = d.exp_value[q];
      if (n > 0) {
        const int off = d.n_nodes[t];
        if (off + n > d.cap) {
          if (lane == 0) { if (!d.overflow[t]) atomicAdd(&d.counters[CNT_FULL], 1ull); d.overflow[t] = 1; d.stalled[t] = 1; }
          have = false;
        } else {
          for (int i = lane; i < n; i += WAVE) {
            const size_t o = base + off + i;
            d.prior[o] = d.exp_score[q * CELLS_PAD + i]; d.visits[o] = 1; d.bsum[o] = 0.f; d.kids_off[o] = -1; d.kids_n[o] = 0;
            d.nmove[o] = d.exp_move[q * CELLS_PAD + i];
          }
          if (lane == 0) { d.kids_off[base + node] = off; d.kids_n[base + node] = (int16_t)n; d.n_nodes[t] = off + n; }
        }
      }
      if (lane == 0) atomicAdd(&d.counters[CNT_EVALS], 1ull);
    }
    __syncthreads();
    if (lane == 0) d.leaf_result[q] = result;
    if (have) have_mask |= 1u << l;
    if (have) {
      for (int j = lane; j < plen; j += WAVE) {
        const size_t o = base + path[j];
        d.visits[o] = d.visits[o] + 1;
        d.bsum[o] = __fadd_rn(d.bsum[o], result);
      }
      if (!prep && lane == 0) atomicAdd(&d.counters[CNT_NONNULL], 1ull);
    }
    for (int j = lane; j < plen; j += WAVE) d.vl[base + path[j]] = 0;
  }
  if (n_lanes > 0 && n_null == n_lanes && lane == 0) d.stalled[t] = 1;
}

// expandAndSimulate (mcts/search.go:259-339) after the network call + the BACKPROPAGATE half of pipeline().
__global__ __launch_bounds__(64) void k_expand(Dev d, GameCfg c, MctsCfg mc, InfDesc inf, int prep, int nl) {
  __shared__ Sh s;
  int g = blockIdx.x, lane = threadIdx.x;
  if (d.ended[g]) return;
  int agent = agent_of(d, g);
  int t = agent * d.G + g;
  const size_t q0 = (size_t)g * d.V;
  const bool use_vl = d.V > 1;
  size_t base = pool_base(d, t, d.cur_pool[t]);
  int n_null = 0, n_lanes = 0;
  unsigned have_mask = 0;   // lanes of this round that backed a value up
  // lanes in order (oracle: MCTS::parallelRound phase 2): expand / follow, Update along the path, undoVirtualLoss
  for (int l = 0; l < nl; l++) {
    const size_t q = q0 + l;
    if (l > 0) { __threadfence(); __syncthreads(); }   // the previous lane's children, visits, sums
    int kind = d.leaf_kind[q];
    if (kind == LEAF_NONE) continue;
    n_lanes++;
    if (!prep && lane == 0) atomicAdd(&d.counters[CNT_SIMS], 1ull);
    const int32_t* path = d.path + q * MAXPATH;
    int plen = d.path_len[q];
    if (kind == LEAF_NULL) {
      n_null++;
      if (use_vl) for (int j = lane; j < plen; j += WAVE) d.vl[base + path[j]] = 0;
      continue;
    }
    int node = path[plen - 1];
    float result = d.leaf_result[q];
    bool have = true;
    bool follower = false;
    if (kind == LEAF_EXPAND && l > 0) {
      // an earlier lane of this round already expanded this very node: the reference's second goroutine arriving while the
      // expansion is in flight — same state, same evaluation, findChild stops duplicate children; the value is backed up again
      for (int l2 = 0; l2 < l && !follower; l2++) {
        const size_t q2 = q0 + l2;
        if (d.leaf_kind[q2] == LEAF_EXPAND && d.path_len[q2] == plen && d.path[q2 * MAXPATH + plen - 1] == node) {
          follower = true;
          result = d.leaf_result[q2];
          have = (have_mask >> l2) & 1u;
        }
      }
    }
    if (kind == LEAF_EXPAND && !follower) {
      int player = d.leaf_player[q];
      const uint8_t* legal = d.leaf_legal + q * CELLS_PAD;
      const int ik = inf.kind[agent];
      const int plen_pol = inf.policy_len[agent];
      const float* pol = nullptr;
      float value = 0.f;
      uint32_t ph = 0;
      if (ik == AGZ_INF_NET) {
        int slot = l * d.G + d.slot_of_game[g];   // lane-major NN slots (k_select)
        pol = inf.policy[agent] + (size_t)slot * plen_pol;
        value = inf.value[agent][slot];
      } else if (ik == AGZ_INF_DUMMY) {  // dummy.go:10-23
        int dp = inf.dummy_player[agent];
        value = dp == 1 ? 1.f : (dp == 2 ? -1.f : 0.f);
      } else if (ik == AGZ_INF_SCRIPT) {  // mcts/example_test.go:40-72 (8 / 9 == 0)
        int mn = d.leaf_ply[q];
        value = (mn == 0 || mn == 1 || mn == 5) ? 0.5f : 0.f;
      } else if (ik == AGZ_INF_HASH) {  // synthetic position hash (oracle/arena.hpp HashNN)
        uint32_t h = 0;
        for (int i = lane; i < c.cells; i += WAVE) h += mix32((uint32_t)i * 4u + (uint32_t)d.leaf_board[q * CELLS_PAD + i] + 1u);
        for (int o = 32; o > 0; o >>= 1) h += __shfl_xor(h, o, 64);
        h += mix32(0xABCD0000u + (uint32_t)player);
        ph = h;
        value = (float)(mix32(h ^ 0xDEADBEEFu) >> 8) * (1.0f / 16777216.0f);
      } else {  // AGZ_INF_UNIFORM, mcts/example_test.go:158-166
        value = 1 / 25.0f;
      }
      auto policy_at = [&](int i) -> float {
        switch (ik) {
          case AGZ_INF_NET: return pol[i];
          case AGZ_INF_DUMMY: return __fdiv_rn(1.f, (float)plen_pol);
          case AGZ_INF_SCRIPT: {
            int mn = d.leaf_ply[q];
            const int8_t cell[9] = {4, 0, 2, 6, 3, 5, 1, 7, 8};
            if (mn >= 0 && mn < 9 && i == cell[mn]) return (mn & 1) ? 0.1f : 0.9f;
            return 0.f;
          }
          case AGZ_INF_HASH: return (float)((mix32(ph + (uint32_t)i * 0x9E3779B9u) >> 8) + 1u) * (1.0f / 16777216.0f);
          default: return 1 / 25.0f;
        }
      };
      if (player == AGZ_WHITE) value = __fsub_rn(1.f, value);  // search.go:278-280
      // nodelist in move order, Pass last (search.go:285-296)
      int n = 0;
      {
        // legality bytes and network priors of all actions first, in one batch of loads (a prior read behind `if (legal)` is a
        // dependent round trip per 64 actions), then the compaction
        constexpr int PER = CELLS_PAD / WAVE;
        uint8_t lg[PER]; float pr[PER];
#pragma unroll
        for (int k = 0; k < PER; k++) {
          const int i = lane + k * WAVE;
          const bool in = i < c.A;
          lg[k] = in ? legal[i] : 0;
          pr[k] = (ik == AGZ_INF_NET && in) ? pol[i] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < PER; k++) {
          const int i = lane + k * WAVE;
          if (k * WAVE >= c.A) break;
          bool ok = lg[k] != 0;
          unsigned long long m = __ballot(ok);
          if (ok) {
            int pos = n + __popcll(m & ((1ull << lane) - 1ull));
            s.fscore[pos] = ik == AGZ_INF_NET ? pr[k] : policy_at(i);
            s.fmove[pos] = i;
          }
          n += __popcll(m);
        }
      }
      if (legal[c.A]) {
        if (lane == 0) { s.fscore[n] = policy_at(plen_pol - 1); s.fmove[n] = AGZ_PASS; }  // passProb = policy[len-1]
        n++;
      }
      __syncthreads();
      // legalSum: sequential float32 sum in list order, as the reference accumulates it
      float legalSum = 0.f;
      if (lane == 0) {   // (four list entries per LDS read; the additions stay one after the other, in list order)
        int i = 0;
        for (; i + 4 <= n; i += 4) {
          const float4 v4 = *reinterpret_cast<const float4*>(&s.fscore[i]);
          legalSum = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(legalSum, v4.x), v4.y), v4.z), v4.w);
        }
        for (; i < n; i++) legalSum = __fadd_rn(legalSum, s.fscore[i]);
      }
      legalSum = __shfl(legalSum, 0, 64);
      if (legalSum > 1.401298464e-45f) {
        for (int i = lane; i < n; i += WAVE) s.fscore[i] = __fdiv_rn(s.fscore[i], legalSum);
      } else {
        float prob = __fdiv_rn(1.f, (float)n);
        for (int i = lane; i < n; i += WAVE) s.fscore[i] = prob;
      }
      __syncthreads();
      result = value;
      if (n > 0) {
        int off = d.n_nodes[t];
        if (off + n > d.cap) {  // pool exhausted: reported, the tree stops growing (AGZ_E_TREE_FULL)
          if (lane == 0) { if (!d.overflow[t]) atomicAdd(&d.counters[CNT_FULL], 1ull); d.overflow[t] = 1; d.stalled[t] = 1; }
          have = false;
        } else {
          // stable descending sort by rank; the sorted list goes through LDS scratch of the board analysis so that the child block is
          // written as whole rows instead of one scattered element per lane and array
          int rnk[RANK_PER];
          stable_desc_ranks(s.fscore, n, lane, rnk);
#pragma unroll
          for (int k = 0; k < RANK_PER; k++) {
            const int i2 = lane + k * WAVE;
            if (i2 < n) { s.touch[rnk[k]] = __float_as_int(s.fscore[i2]); s.ghash[rnk[k]] = s.fmove[i2]; }
          }
          __syncthreads();
          for (int r = lane; r < n; r += WAVE) {
            size_t o = base + off + r;
            d.prior[o] = __int_as_float(s.touch[r]); d.visits[o] = 1; d.bsum[o] = 0.f; d.kids_off[o] = -1; d.kids_n[o] = 0;  // tree.go:106-117
            d.nmove[o] = (int16_t)s.ghash[r];
          }
          if (lane == 0) { d.kids_off[base + node] = off; d.kids_n[base + node] = (int16_t)n; d.n_nodes[t] = off + n; }
        }
      }
      if (lane == 0) atomicAdd(&d.counters[CNT_EVALS], 1ull);
    }
    __syncthreads();
    if (lane == 0) d.leaf_result[q] = result;   // what this lane backs up (read by followers)
    if (have) have_mask |= 1u << l;
    if (have) {  // Update along the path (search.go:251-253, node.go:70-76): same black-perspective value at every level
      for (int j = lane; j < plen; j += WAVE) {
        size_t o = base + path[j];
        d.visits[o] = d.visits[o] + 1;
        d.bsum[o] = __fadd_rn(d.bsum[o], result);
      }
      if (!prep && lane == 0) atomicAdd(&d.counters[CNT_NONNULL], 1ull);
    }
    if (use_vl) for (int j = lane; j < plen; j += WAVE) d.vl[base + path[j]] = 0;   // undoVirtualLoss (search.go:254)
  }
  // a round of nothing but null simulations repeats forever in a deterministic search
  if (n_lanes > 0 && n_null == n_lanes && lane == 0) d.stalled[t] = 1;
}

// bestMove + Policies + Arena.Play's per-move bookkeeping (search.go:341-390,152-161; arena.go:98-138)
// forced != nullptr: agz_arena_apply_moves — the move of every unfinished game comes from outside (no search, no
// example, the mover's tree is left alone and re-roots over the extra plies at its next search).
// search_only != 0: MCTS.Search's own tail only (search.go:151-163) — bestMove, t.prev = current.Clone(), cachedPolicies++ —
// the move goes to d.best_out[g] and the game is NOT advanced (the caller of Search applies it to its own game.State).
__global__ __launch_bounds__(64) void k_end_move(Dev d, GameCfg c, MctsCfg mc, int record, int restart, const int32_t* forced,
                                                 int search_only) {
  __shared__ Sh s;
  __shared__ uint32_t cv[CELLS_PAD];   // child visits
  __shared__ int16_t ckn[CELLS_PAD];   // child kids_n
  __shared__ int32_t rankv[CELLS_PAD];
  int g = blockIdx.x, lane = threadIdx.x;
  if (d.ended[g]) return;
  int agent = agent_of(d, g);
  int t = agent * d.G + g;
  int player = d.to_move[g];
  const bool use_ring = true;
  St st;
  load_state(c, d, g, s, st, use_ring, lane);
  size_t base = pool_base(d, t, d.cur_pool[t]);
  int off = d.kids_off[base], n = off >= 0 ? d.kids_n[base] : 0;
  int best = AGZ_PASS;
  float bestScore = 0.f;
  const bool is_forced = forced != nullptr;
  if (is_forced) {
    n = 0; record = 0;
    best = forced[g];
    if (best == AGZ_NO_MOVE) return;   // this game is not advanced by the call
    // State.Check (game/state.go:136): Resign always ends the game; Pass where the game has one; a board move must be legal
    bool legal;
    if (best == AGZ_RESIGN) legal = true;
    else if (best == AGZ_PASS) legal = c.pass_legal;
    else if (best < 0 || best >= c.A) legal = false;
    else if (c.go_like) { analyse(c, s, nullptr, lane, false); legal = go_legal(c, s, best, player); }
    else legal = s.board[best] == AGZ_NONE;
    if (!legal) {
      if (lane == 0) atomicAdd(&d.counters[CNT_ILLEGAL], 1ull);
      return;
    }
  }
  if (n > 0) {
    // children -> LDS: fscore=prior, cv=visits, touch=bsum bits, fmove=move, gsize=kids_off, ckn=kids_n
    for (int i = lane; i < n; i += WAVE) {
      s.fscore[i] = d.prior[base + off + i]; cv[i] = d.visits[base + off + i];
      s.touch[i] = __float_as_int(d.bsum[base + off + i]); s.fmove[i] = d.nmove[base + off + i];
      s.gsize[i] = d.kids_off[base + off + i]; ckn[i] = d.kids_n[base + off + i];
    }
    __syncthreads();
    // fancySort (utils.go:18-47), stable
    for (int i = lane; i < n; i += WAVE) {
      uint32_t vi = cv[i];
      float ei = evaluate(__int_as_float(s.touch[i]), vi, player), pi = s.fscore[i];
      int rank = 0;
      for (int j = 0; j < n; j++) {
        uint32_t vj = cv[j];
        bool jl, il;  // less(j,i), less(i,j)
        if (vj != vi) { jl = vj > vi; il = vi > vj; }
        else if (vi == 0) { float pj = s.fscore[j]; jl = pj > pi; il = pi > pj; }
        else { float ej = evaluate(__int_as_float(s.touch[j]), vj, player); jl = ej > ei; il = ei > ej; }
        rank += jl || (!jl && !il && j < i);
      }
      rankv[i] = rank;
    }
    __syncthreads();
    // label[] := order (sorted position -> original index)
    for (int i = lane; i < n; i += WAVE) s.label[rankv[i]] = i;
    __syncthreads();
    // randomizeChildren (tree.go:212-247) when moveNum < RandomCount
    if (move_number(c, st.ply) < mc.RandomCount) {
      if (lane == 0) {
        float accum = 0.f, norm = 0.f;
        int nacc = 0, index = 0;
        bool bail = false;
        for (int q = 0; q < n && !bail; q++) {
          uint32_t v = cv[s.label[q]];
          if (norm == 0.f) { norm = (float)v; if (v <= mc.RandomMinVisits) bail = true; }
          if (!bail && v > mc.RandomMinVisits) {
            float ex = __fdiv_rn(1.f, mc.RandomTemperature), rt = __fdiv_rn((float)v, norm);
            // math32.Pow(x, y) = float32(math.Pow(float64(x), float64(y))) (tree.go:228): the power in DOUBLE, rounded once (powf differs in the last place)
            accum = __fadd_rn(accum, ex == 1.f ? rt : (float)pow((double)rt, (double)ex));  // pow(x, 1) == x exactly
            s.libs[nacc++] = __float_as_int(accum);
          }
        }
        if (!bail) {
          unsigned long long z = (d.rng[t] += 0x9E3779B97F4A7C15ull);
          z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
          float rnd = __fmul_rn((float)(z >> 40) * (1.0f / 16777216.0f), accum);
          for (int q = 0; q < nacc; q++) if (rnd < __int_as_float(s.libs[q])) { index = q; break; }
          if (index != 0)
            for (int q = 0; q < n - index; q++) { int tmp = s.label[q]; s.label[q] = s.label[q + index]; s.label[q + index] = tmp; }
        }
      }
      __syncthreads();
    }
    // write the block back in its new order (the reference sorts children[root] in place)
    for (int q = lane; q < n; q += WAVE) {
      int i = s.label[q];
      size_t o = base + off + q;
      d.prior[o] = s.fscore[i]; d.visits[o] = cv[i]; d.bsum[o] = __int_as_float(s.touch[i]);
      d.nmove[o] = (int16_t)s.fmove[i]; d.kids_off[o] = s.gsize[i]; d.kids_n[o] = ckn[i];
    }
    int f = s.label[0];
    best = s.fmove[f];
    bestScore = evaluate(__int_as_float(s.touch[f]), cv[f], player);
    // first non-pass child in sorted order (noPass, search.go:530-545; children are legal moves of this state)
    int np = 0x7fffffff;
    for (int q = lane; q < n; q += WAVE) if (s.fmove[s.label[q]] != AGZ_PASS && q < np) np = q;
    for (int o = 32; o > 0; o >>= 1) { int f2 = __shfl_xor(np, o, 64); np = f2 < np ? f2 : np; }
    bool do_nopass = false;
    float rootScore = d.prior[base];
    bool losing = (rootScore > 0 && player == AGZ_WHITE) || (rootScore < 0 && player == AGZ_BLACK);
    int lastmv = d.last_move[g];
    if (mc.PassPreference == AGZ_DONT_PREFER_PASS && best == AGZ_PASS) do_nopass = true;
    else if (!mc.DumbPass && best == AGZ_PASS) { if (losing) do_nopass = true; }
    else if (!mc.DumbPass && lastmv == AGZ_PASS) { if (!losing) best = AGZ_PASS; }  // LastMove() of an empty history IS Pass (-1)
    if (do_nopass && np != 0x7fffffff) {
      int i = s.label[np];
      best = s.fmove[i];
      bestScore = 1.f;
      if (cv[i] != 0) bestScore = evaluate(__int_as_float(s.touch[i]), cv[i], player);
    }
    // shouldResign (search.go:502-528)
    if (best == AGZ_PASS && mc.PassPreference != AGZ_DONT_RESIGN && mc.ResignPercentage != 0.f) {
      int threshold = (mc.maxDepth) / 4;
      if (move_number(c, st.ply) > threshold) {
        float rt = mc.ResignPercentage < 0.f ? 0.1f : mc.ResignPercentage;
        if (!(bestScore > rt)) best = AGZ_RESIGN;
      }
    }
  }
  __syncthreads();
  // board hash of the searched state (search.go:96): zobrist for komi/wq, FNV for mnk/c4
  uint32_t hash = c.go_like ? st.hash : fnv_board(c, s.board);
  int pcn = d.pc_n[t];
  if (n > 0 && lane == 0) {  // t.cachedPolicies[sa{boardHash, retVal}]++
    d.pc_hash[(size_t)t * d.moves_stride + pcn] = hash; d.pc_move[(size_t)t * d.moves_stride + pcn] = (int16_t)best; d.pc_n[t] = pcn + 1;
  }
  if (n > 0) pcn++;
  __syncthreads();
  if (record) {  // arena.go:105-123
    // Policies (tree.go:128-142): counts of (hash, a) for a in [0, A], normalised; NaN when the move was Pass/Resign
    float tot = 0.f;
    for (int q = 0; q < pcn; q++) {
      int mv = d.pc_move[(size_t)t * d.moves_stride + q];
      if (d.pc_hash[(size_t)t * d.moves_stride + q] == hash && mv >= 0 && mv <= c.A) tot += 1.f;
    }
    if (tot > 0.f) {  // validPolicies (arena.go:241-251): sum 0 -> NaN -> dropped
      int e = -1;
      if (lane == 0) e = atomicAdd(d.ex_count, 1);
      e = __shfl(e, 0, 64);
      if (e < d.ex_cap) {
        float* P = d.ex_policy + (size_t)e * (c.A + 1);
        for (int a = lane; a <= c.A; a += WAVE) {
          float cnt = 0.f;
          for (int q = 0; q < pcn; q++)
            if (d.pc_hash[(size_t)t * d.moves_stride + q] == hash && d.pc_move[(size_t)t * d.moves_stride + q] == a) cnt += 1.f;
          P[a] = __fdiv_rn(cnt, tot);
        }
        encode_nchw(c, s, st, d.ex_planes + (size_t)e * c.F * c.cells, lane);
        if (lane == 0) {
          d.ex_value[e] = (float)player;  // "THIS IS A HACK": mover colour until the game ends (arena.go:111-113)
          d.ex_labelled[e] = 0;
          d.ex_game[e] = g; d.ex_prev[e] = d.ex_last[g]; d.ex_last[g] = e;
          atomicAdd(&d.counters[CNT_EXAMPLES], 1ull);
        }
      } else if (lane == 0) {   // buffer full: the example is lost — counted (agz_arena_stats.examples_dropped)
        atomicSub(d.ex_count, 1);
        atomicAdd(&d.counters[CNT_DROPPED], 1ull);
      }
    }
  }
  // t.prev = t.current.Clone() (search.go:152)
  if (!is_forced)
    for (int i = lane; i < c.cells; i += WAVE) d.prev_board[(size_t)t * CELLS_PAD + i] = s.board[i];
  if (lane == 0 && n > 0) { d.prev_ply[t] = st.ply; d.has_prev[t] = 1; }
  __syncthreads();
  if (search_only) {
    if (lane == 0) d.best_out[g] = best;
    return;
  }
  // a.game = a.game.Apply(PlayerMove{player, best}) (arena.go:127-130)
  int ended = 0, winner = AGZ_NONE;
  int pass_count = d.pass_count[g];
  pass_count = (best == AGZ_PASS) ? pass_count + 1 : 0;
  float capb = d.cap_b[g], capw = d.cap_w[g];
  if (lane == 0) {
    d.moves[(size_t)g * d.moves_stride + st.ply] = (int16_t)best;
    int na = d.n_amoves[g];
    if (na < d.moves_stride) { d.amoves[(size_t)g * d.moves_stride + na] = (int16_t)best; d.n_amoves[g] = na + 1; }
  }
  bool resigned = best == AGZ_RESIGN;
  if (!resigned) {
    bool legal_pass = c.pass_legal;
    if (best == AGZ_PASS && !legal_pass) {
      // mnk/komi Apply of an illegal pass leaves the state unchanged (mnk.go:123-125, komi/game.go:107-110)
    } else {
      st.to_move = player;
      int taken = apply_move(c, d, s, st, best, use_ring, lane);
      if (!c.flip_in_tree) st.ply = st.ply;  // c4: history still grows (game.go:60-64); MoveNumber() stays 1
      if (player == AGZ_BLACK) capb += (float)taken; else capw += (float)taken;
    }
  }
  // switchPlayer + (next Search) SetToMove: the next agent's colour
  int next_colour = opp(player);
  // Ended(): per game
  if (resigned) { ended = 1; winner = opp(player); }
  else if (pass_count >= 2) {  // arena.go:135 breaks without re-evaluating Ended(); wq (completed) scores the board
    ended = 1;
    if (c.kind == AGZ_GAME_WQ) {
      analyse(c, s, nullptr, lane);
      float b, w; area_scores(c, s, lane, &b, &w);
      w = __fadd_rn(w, c.komi);
      winner = (w == b) ? AGZ_NONE : (w > b ? AGZ_WHITE : AGZ_BLACK);
    }
  } else if (c.max_moves > 0 && st.ply >= c.max_moves) {
    ended = 1;
    if (c.kind == AGZ_GAME_WQ) {
      analyse(c, s, nullptr, lane);
      float b, w; area_scores(c, s, lane, &b, &w);
      w = __fadd_rn(w, c.komi);
      winner = (b > w) ? AGZ_BLACK : (w > b ? AGZ_WHITE : AGZ_NONE);
    }
  } else if (c.kind == AGZ_GAME_MNK) {  // mnk.go:161-174
    if (mnk_is_winner(c, s.board, AGZ_BLACK)) { ended = 1; winner = AGZ_BLACK; }
    else if (mnk_is_winner(c, s.board, AGZ_WHITE)) { ended = 1; winner = AGZ_WHITE; }
    else { int e = 0; for (int i = lane; i < c.cells; i += WAVE) e |= (s.board[i] == AGZ_NONE); ended = __syncthreads_or(e) ? 0 : 1; }
  } else if (c.kind == AGZ_GAME_C4) {  // c4/game.go:164-183 (passCount of the GAME: consecutive passes > 2)
    int w = c4_check_win(c, s.board);
    if (w != AGZ_NONE) { ended = 1; winner = w; }
    else { int e = 0; for (int i = lane; i < c.cells; i += WAVE) e |= (s.board[i] == AGZ_NONE); ended = __syncthreads_or(e) ? 0 : 1; }
  } else if (c.kind == AGZ_GAME_KOMI) {  // komi/game.go:145-187
    float kf = (float)c.k;
    if (capw >= kf) { ended = 1; winner = AGZ_WHITE; }
    else if (capb >= kf) { ended = 1; winner = AGZ_BLACK; }
    else {
      analyse(c, s, nullptr, lane, false);
      int cur = 0, op = 0;
      for (int i = lane; i < c.cells; i += WAVE) { cur |= go_legal(c, s, i, next_colour); op |= go_legal(c, s, i, opp(next_colour)); }
      cur = __syncthreads_or(cur); op = __syncthreads_or(op);
      if (!(cur && op)) { ended = 1; winner = capw > capb ? AGZ_WHITE : (capb > capw ? AGZ_BLACK : AGZ_NONE); }
    }
  } else {  // wq: passes >= 2 ends (game.go:94-115)
    if (st.passes >= 2) {
      ended = 1;
      analyse(c, s, nullptr, lane);
      float b, w; area_scores(c, s, lane, &b, &w);
      w = __fadd_rn(w, c.komi);
      winner = (w == b) ? AGZ_NONE : (w > b ? AGZ_WHITE : AGZ_BLACK);
    }
  }
  __syncthreads();
  // write the state back
  for (int i = lane; i < c.cells; i += WAVE) d.board[(size_t)g * CELLS_PAD + i] = s.board[i];
  {
    int slot = st.ply % RING;
    for (int i = lane; i < c.cells; i += WAVE) d.ring[((size_t)g * RING + slot) * CELLS_PAD + i] = s.ring[slot][i];
  }
  if (lane == 0) {
    d.to_move[g] = next_colour; d.ply[g] = st.ply; d.passes[g] = st.passes; d.zhash[g] = st.hash;
    d.pass_count[g] = pass_count; d.cap_b[g] = capb; d.cap_w[g] = capw; d.last_move[g] = best;
    d.ended[g] = ended; d.winner[g] = winner;
    atomicAdd(&d.counters[CNT_MOVES], 1ull);
    if (ended) {
      atomicAdd(&d.counters[CNT_GAMES], 1ull);
      {  // Agent.Wins / Loss / Draw (arena.go:156-171)
        int a_colour = d.a_is_black[g] ? AGZ_BLACK : AGZ_WHITE;
        atomicAdd(&d.counters[winner == AGZ_NONE ? CNT_DRAWS : (winner == a_colour ? CNT_A_WINS : CNT_B_WINS)], 1ull);
      }
      // label the game's examples (arena.go:146-155)
      for (int e = d.ex_last[g]; e >= 0; e = d.ex_prev[e]) {
        float mover = d.ex_value[e];
        d.ex_value[e] = winner == AGZ_NONE ? 0.f : (mover == (float)winner ? 1.f : -1.f);
        d.ex_labelled[e] = 1;
      }
      d.ex_last[g] = -1;
    }
  }
  // continuous self-play (AZ.SelfPlay in a loop, agogo.go:110-114): a finished game is replaced at once by a new
  // one — fresh board, fresh trees (arena.go:175-176 builds new mcts.New trees per game), colours drawn again
  if (ended && restart) {
    __syncthreads();
    for (int i = lane; i < CELLS_PAD; i += WAVE) d.board[(size_t)g * CELLS_PAD + i] = 0;
    for (int i = lane; i < RING * CELLS_PAD; i += WAVE) d.ring[(size_t)g * RING * CELLS_PAD + i] = 0;
    if (lane == 0) {
      unsigned long long z = (d.rng_game[g] += 0x9E3779B97F4A7C15ull);
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
      d.a_is_black[g] = (z >> 63) == 0 ? 1 : 0;
      d.to_move[g] = AGZ_BLACK; d.ply[g] = 0; d.passes[g] = 0; d.pass_count[g] = 0; d.ended[g] = 0; d.winner[g] = AGZ_NONE;
      d.last_move[g] = AGZ_PASS; d.cap_b[g] = 0.f; d.cap_w[g] = 0.f; d.zhash[g] = 0; d.n_amoves[g] = 0; d.hist_from[g] = 0;
      for (int ag = 0; ag < 2; ag++) {
        int tt = ag * d.G + g;
        d.n_nodes[tt] = 0; d.cur_pool[tt] = 0; d.has_root[tt] = 0; d.has_prev[tt] = 0; d.prev_ply[tt] = 0; d.stalled[tt] = 0;
        d.pc_n[tt] = 0;
        d.rng[tt] = d.rng_game[g] * 2ull + 1ull + (unsigned long long)ag;
      }
    }
  }
}

// Synthetic openings ("play u uniformly-random legal moves from the empty board", the benchmark's board
// batches): every unfinished game with remaining[g] > 0 draws the (z mod n_legal)-th legal board move in ascending cell
// order for the player to move, z = SplitMix64 finaliser of (seed, game, arena move count); without a legal board move:
// Pass where the game has one, else the game is left alone.  The move goes to forced[g] and is applied by k_end_move's
// forced path (State.Check + Apply + Ended, no search, no example).  oracle: orc_arena_random_move.
__global__ __launch_bounds__(64) void k_random_pick(Dev d, GameCfg c, int32_t* remaining, unsigned long long seed, int32_t* forced) {
  __shared__ Sh s;
  const int g = blockIdx.x, lane = threadIdx.x;
  if (d.ended[g] || remaining[g] <= 0) { if (lane == 0) forced[g] = AGZ_NO_MOVE; return; }
  St st;
  load_state(c, d, g, s, st, false, lane);
  const int player = st.to_move;
  if (c.go_like) analyse(c, s, nullptr, lane, false);
  int total = 0;
  for (int b0 = 0; b0 < c.A; b0 += WAVE) {
    const int i = b0 + lane;
    const bool ok = i < c.A && (c.go_like ? go_legal(c, s, i, player) : s.board[i] == AGZ_NONE);
    total += __popcll(__ballot(ok));
  }
  int mv = AGZ_NO_MOVE;
  if (total == 0) {
    if (c.pass_legal) mv = AGZ_PASS;
  } else {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(g + 1) +
                           0xD1B54A32D192ED03ull * (unsigned long long)(d.n_amoves[g] + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    const int want = (int)(z % (unsigned long long)total);
    int seen = 0;
    for (int b0 = 0; b0 < c.A; b0 += WAVE) {
      const int i = b0 + lane;
      const bool ok = i < c.A && (c.go_like ? go_legal(c, s, i, player) : s.board[i] == AGZ_NONE);
      const unsigned long long m = __ballot(ok);
      const int cnt = __popcll(m);
      if (want >= seen && want < seen + cnt) {
        // the (want - seen)-th set bit of m
        unsigned long long mm = m;
        for (int q = 0; q < want - seen; q++) mm &= mm - 1;
        mv = b0 + (__ffsll((long long)mm) - 1);
      }
      seen += cnt;
    }
  }
  if (lane == 0) { forced[g] = mv; remaining[g] = mv == AGZ_NO_MOVE ? 0 : remaining[g] - 1; }
}

}  // namespace agz

using namespace agz;

// ---------------------------------------------------------------------------------------------------
struct agz_arena {
  agz_ctx* ctx = nullptr;
  GameCfg gc{};
  MctsCfg mc{};
  Dev d{};
  int G = 0;
  uint64_t seed = 0, seed0 = 0;  // seed advances per reset (colour draws); seed0 fixes the per-tree RNG streams
  std::vector<void*> allocs;
  int inf_kind[2] = {AGZ_INF_DUMMY, AGZ_INF_DUMMY};
  agz_net* net[2] = {nullptr, nullptr};
  float* d_policy[2] = {nullptr, nullptr};
  float* d_value[2] = {nullptr, nullptr};
  std::vector<uint8_t> a_is_black;
  int32_t* d_forced = nullptr;   // agz_arena_apply_moves staging
  std::vector<int32_t> slot_host;
  int32_t* d_remaining = nullptr;   // agz_arena_random_moves staging
  unsigned long long prep_expand_seen = 0;   // CNT_PREP_EXPAND at the last prepareRoot step
  bool prep_compact = true;                  // agz_arena_set_prep_compact (agz_debug.h)
  int last_prep_batch = 0, last_prep_roots = 0;   // boards in / roots evaluated by the last prepareRoot forward (0: skipped)
  int nA_slots = 0, nB_slots = 0;
  bool in_move = false;
  bool restart = false;  // continuous self-play: finished games restart immediately
  int moves_done = 0;
  // AGZ_INF_CALLBACK (mcts.Inferencer as a host function, mcts/mcts.go:15-18): the callee, its policy width, page-locked staging
  agz_infer_fn cb_fn[2] = {nullptr, nullptr};
  void* cb_user[2] = {nullptr, nullptr};
  int cb_plen[2] = {0, 0};
  std::vector<void*> host_allocs;
  int cb_rows = 0;                                   // G * V the staging below was sized for
  int32_t *h_kind = nullptr, *h_player = nullptr, *h_ply = nullptr, *h_tomove = nullptr, *h_ab = nullptr, *h_ended = nullptr;
  int8_t* h_lboard = nullptr;
  float* h_planes = nullptr;                         // [G*V][F*cells] as the device wrote them
  float *h_pk_planes = nullptr, *h_pk_policy = nullptr, *h_pk_value = nullptr;   // the packed batch handed to the callee
  int32_t *h_pk_board = nullptr, *h_pk_tomove = nullptr, *h_pk_mn = nullptr, *h_pk_game = nullptr;
  float* h_policy[2] = {nullptr, nullptr};           // [G*V][policy_len] in NN-slot order, uploaded for k_expand
  float* h_value[2] = {nullptr, nullptr};
  int64_t cb_calls = 0, cb_leaves = 0;
  bool pool_grow = false;   // AGZ_POOL_GROW: before every search the pools are made large enough for it (ensure_pool_room)
  long long room_sims = 0, sims_this_move = 0;   // simulations the current capacity is guaranteed for / enqueued since begin_move
  int pool_grows = 0;       // how often the pools were re-allocated
  int ensure_pool_room(long long sims);
  bool pool_stop = false;   // agz_arena_set_pool_policy: a full node pool stops that tree's search for the move (the reference's MAXTREESIZE rule) instead of failing the call
  size_t h_pk_policy_cap = 0, h_policy_cap[2] = {0, 0}, h_value_cap[2] = {0, 0};

  template <typename T>
  int alloc(T** p, size_t n) {
    void* q = nullptr;
    AGZ_HIP_TRY(hipMalloc(&q, n * sizeof(T)));
    AGZ_HIP_TRY(hipMemsetAsync(q, 0, n * sizeof(T), ctx->stream));
    allocs.push_back(q);
    *p = (T*)q;
    return AGZ_OK;
  }
  bool split_nets() const { return inf_kind[0] == AGZ_INF_NET && inf_kind[1] == AGZ_INF_NET && net[0] != net[1]; }
  template <typename T>
  int halloc(T** p, size_t n) {
    void* q = nullptr;
    AGZ_HIP_TRY(hipHostMalloc(&q, std::max<size_t>(n, 1) * sizeof(T), hipHostMallocDefault));
    host_allocs.push_back(q);
    *p = (T*)q;
    return AGZ_OK;
  }
  int cb_setup();                    // (re)allocate the callback staging for G * V leaves
  int cb_run(int nl);                // leaves of callback agents -> host function -> d_policy / d_value
  int update_slots();
  int nn_step(int prep, int nl = 1);   // nl lanes per tree in this round (<= d.V)
};

// prepareRoot evaluates only roots without children (search.go:392-408): with tree reuse a minority of an arena's games (19x19, 512
// games: 80-130 per move boundary — the opponent's reply was never expanded in this agent's tree).  Their input planes, written by
// k_select at the games' own slots, are packed to the front of the batch IN PLACE: game g's planes move to slot rank(g) = the number
// of earlier games that need the network.  Thread i moves float4 i of every board, in game order: slot rank(g) <= slot g, and a
// slot is overwritten only after the same thread has moved its element out — no barrier, one pass, deterministic.
__global__ __launch_bounds__(256) void k_prep_compact(Dev d, float* act, int slot_f4) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float4* a4 = reinterpret_cast<float4*>(act);
  int rank = 0;
  for (int g = 0; g < d.G; g++) {
    const bool need = d.leaf_kind[(size_t)g * d.V] == LEAF_EXPAND && !d.ended[g];
    if (i == 0) d.prep_slot[g] = need ? rank : 0;
    if (!need) continue;
    const int src = d.slot_of_game[g];
    if (src != rank && i < slot_f4) a4[(size_t)rank * slot_f4 + i] = a4[(size_t)src * slot_f4 + i];
    rank++;
  }
}

// ---- AGZ_POOL_GROW -----------------------------------------------------------------------------------------------------------------
// Node pools that cannot overflow.  A search of `sims` simulations adds at most (sims + 1) expansions of at most A + 1 children each to a
// tree (prepareRoot's included); before the search the host reads every tree's node count (T words, one synchronisation per move) and, if the
// fullest tree could outgrow the pool, re-allocates ALL pools at a larger capacity (trees keep one stride: pool_base) and copies every
// tree's live pool over — node indices, child offsets and therefore every result stay what they were.  Rare (a narrow tree that keeps most of
// its nodes move after move), tens of milliseconds when it happens; memory follows the fullest tree.
__global__ __launch_bounds__(256) void k_pool_copy(Dev o, Dev n) {
  const int t = blockIdx.y;
  const int pool = o.cur_pool[t], cnt = o.n_nodes[t];
  const size_t so = ((size_t)t * 2 + pool) * o.cap, sn = ((size_t)t * 2 + pool) * n.cap;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < cnt; i += gridDim.x * 256) {
    n.prior[sn + i] = o.prior[so + i]; n.visits[sn + i] = o.visits[so + i]; n.bsum[sn + i] = o.bsum[so + i];
    n.kids_off[sn + i] = o.kids_off[so + i]; n.kids_n[sn + i] = o.kids_n[so + i]; n.nmove[sn + i] = o.nmove[so + i];
  }
}

int agz_arena::ensure_pool_room(long long sims) {
  hipStream_t s = ctx->stream;
  const int T = d.T;
  std::vector<int32_t> nn(T);
  AGZ_HIP_TRY(hipMemcpyAsync(nn.data(), d.n_nodes, (size_t)T * 4, hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipStreamSynchronize(s));
  long long most = 0;
  for (int t = 0; t < T; t++) most = std::max<long long>(most, nn[t]);
  const long long need = most + (sims + 2) * (long long)(gc.A + 1) + 16;
  if (need <= d.cap) return AGZ_OK;
  long long ncap = std::max(need + need / 2, 2ll * d.cap);
  size_t free_b = 0, total_b = 0;
  AGZ_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
  const size_t per_node = 4 + 4 + 4 + 4 + 2 + 2 + 1;
  auto bytes = [&](long long c) { return (size_t)T * 2 * (size_t)c * per_node; };
  if (bytes(ncap) + (1ull << 30) > free_b) ncap = need;                      // (the old pools stay allocated until the copy is done)
  AGZ_REQUIRE(ncap < (1ll << 30) && bytes(ncap) + (256ull << 20) <= free_b, AGZ_E_NOMEM,
              "AGZ_POOL_GROW: the fullest tree holds %lld nodes, the next search may need %lld per pool — %zu MB for %d trees do not fit the free device memory (%zu MB)",
              most, need, bytes(need) >> 20, T, free_b >> 20);
  Dev nd = d;
  nd.cap = (int)ncap;
  const size_t pool = (size_t)T * 2 * (size_t)ncap;
  void* old[7] = {d.prior, d.visits, d.bsum, d.kids_off, d.kids_n, d.nmove, d.vl};
  int r;
  if ((r = alloc(&nd.prior, pool)) != AGZ_OK || (r = alloc(&nd.visits, pool)) != AGZ_OK || (r = alloc(&nd.bsum, pool)) != AGZ_OK ||
      (r = alloc(&nd.kids_off, pool)) != AGZ_OK || (r = alloc(&nd.kids_n, pool)) != AGZ_OK || (r = alloc(&nd.nmove, pool)) != AGZ_OK ||
      (r = alloc(&nd.vl, pool)) != AGZ_OK)
    return r;                                                                  // (whatever was allocated is on the arena's list and goes with it)
  hipLaunchKernelGGL(k_pool_copy, dim3(64, T), dim3(256), 0, s, d, nd);
  AGZ_HIP_TRY(hipStreamSynchronize(s));
  for (void* p : old) {
    allocs.erase(std::remove(allocs.begin(), allocs.end(), p), allocs.end());
    hipFree(p);
  }
  d = nd;
  pool_grows++;
  return AGZ_OK;
}

// ---- AGZ_INF_CALLBACK ------------------------------------------------------------------------------------------------------------
// mcts.New(game, conf, nn Inferencer) takes ANY implementation of `Infer(state) (policy, value)` (mcts/mcts.go:15-18, tree.go:80).  The
// device search meets such an inferencer between its two kernels: k_select leaves every expandable leaf's encoded planes (NCHW, the
// arena's encoder), board, mover and move number; the leaves of the callback agents are packed into ONE batch, the host function fills
// policy [n][policy_len] and value [n], and k_expand consumes the rows exactly as it consumes a network's.  One host round trip per
// simulation step (all games at once): the boundary for caller-supplied networks and a network-independent differential hook — not
// the measured path (AGZ_INF_NET keeps everything on the device).
int agz_arena::cb_setup() {
  const int rows = G * d.V;
  const size_t pl = (size_t)gc.F * gc.cells;
  if (!d.cb_planes || cb_rows != rows) {
    int r;
    // (the previous, smaller staging stays on the allocation lists and is released with the arena, as agz_arena_set_parallel's scratch)
    if ((r = alloc(&d.cb_planes, (size_t)rows * pl)) != AGZ_OK) return r;
    if ((r = halloc(&h_kind, rows)) != AGZ_OK || (r = halloc(&h_player, rows)) != AGZ_OK || (r = halloc(&h_ply, rows)) != AGZ_OK ||
        (r = halloc(&h_tomove, G)) != AGZ_OK || (r = halloc(&h_ab, G)) != AGZ_OK || (r = halloc(&h_ended, G)) != AGZ_OK ||
        (r = halloc(&h_lboard, (size_t)rows * CELLS_PAD)) != AGZ_OK || (r = halloc(&h_planes, (size_t)rows * pl)) != AGZ_OK ||
        (r = halloc(&h_pk_planes, (size_t)rows * pl)) != AGZ_OK || (r = halloc(&h_pk_value, rows)) != AGZ_OK ||
        (r = halloc(&h_pk_board, (size_t)rows * gc.cells)) != AGZ_OK || (r = halloc(&h_pk_tomove, rows)) != AGZ_OK ||
        (r = halloc(&h_pk_mn, rows)) != AGZ_OK || (r = halloc(&h_pk_game, rows)) != AGZ_OK)
      return r;
    cb_rows = rows;
    h_pk_policy_cap = 0; h_policy_cap[0] = h_policy_cap[1] = 0; h_value_cap[0] = h_value_cap[1] = 0;   // (rows changed: size everything anew)
  }
  // (the row buffers grow only: a caller that re-registers its inferencer again and again does not accumulate page-locked memory)
  const size_t pk_need = (size_t)rows * std::max(std::max(cb_plen[0], cb_plen[1]), 1);
  if (pk_need > h_pk_policy_cap) { int r = halloc(&h_pk_policy, pk_need); if (r != AGZ_OK) return r; h_pk_policy_cap = pk_need; }
  for (int a = 0; a < 2; a++) {
    if (inf_kind[a] != AGZ_INF_CALLBACK) continue;
    const size_t need = (size_t)rows * cb_plen[a];
    if (d_policy[a]) { hipFree(d_policy[a]); d_policy[a] = nullptr; }
    if (d_value[a]) { hipFree(d_value[a]); d_value[a] = nullptr; }
    AGZ_HIP_TRY(hipMalloc(&d_policy[a], need * sizeof(float)));
    AGZ_HIP_TRY(hipMalloc(&d_value[a], (size_t)rows * sizeof(float)));
    if (need > h_policy_cap[a] || (size_t)rows > h_value_cap[a]) {
      int r;
      if ((r = halloc(&h_policy[a], need)) != AGZ_OK || (r = halloc(&h_value[a], rows)) != AGZ_OK) return r;
      h_policy_cap[a] = need; h_value_cap[a] = (size_t)rows;
    }
    memset(h_policy[a], 0, need * sizeof(float));
    memset(h_value[a], 0, (size_t)rows * sizeof(float));
  }
  return AGZ_OK;
}

int agz_arena::cb_run(int nl) {
  hipStream_t s = ctx->stream;
  const int rows = G * d.V;
  const size_t pl = (size_t)gc.F * gc.cells;
  AGZ_HIP_TRY(hipMemcpyAsync(h_kind, d.leaf_kind, (size_t)rows * 4, hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipMemcpyAsync(h_player, d.leaf_player, (size_t)rows * 4, hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipMemcpyAsync(h_ply, d.leaf_ply, (size_t)rows * 4, hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipMemcpyAsync(h_tomove, d.to_move, (size_t)G * 4, hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipMemcpyAsync(h_ab, d.a_is_black, (size_t)G * 4, hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipMemcpyAsync(h_ended, d.ended, (size_t)G * 4, hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipMemcpyAsync(h_lboard, d.leaf_board, (size_t)rows * CELLS_PAD, hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipMemcpyAsync(h_planes, d.cb_planes, (size_t)rows * pl * 4, hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipStreamSynchronize(s));
  for (int a = 0; a < 2; a++) {
    if (inf_kind[a] != AGZ_INF_CALLBACK) continue;
    if (a == 1 && inf_kind[0] == AGZ_INF_CALLBACK && cb_fn[0] == cb_fn[1] && cb_user[0] == cb_user[1] && cb_plen[0] == cb_plen[1]) continue;   // one callee for both agents: served below in ONE call
    const bool both = a == 0 && inf_kind[1] == AGZ_INF_CALLBACK && cb_fn[0] == cb_fn[1] && cb_user[0] == cb_user[1] && cb_plen[0] == cb_plen[1];
    const int plen = cb_plen[a];
    int n = 0;
    std::vector<int> qs, ags;
    for (int g = 0; g < G; g++) {
      if (h_ended[g]) continue;
      const int agent = ((h_tomove[g] == AGZ_BLACK) == (h_ab[g] != 0)) ? 0 : 1;
      if (agent != a && !(both && agent == 1)) continue;
      for (int l = 0; l < nl; l++) {
        const int q = g * d.V + l;
        if (h_kind[q] != LEAF_EXPAND) continue;
        memcpy(h_pk_planes + (size_t)n * pl, h_planes + (size_t)q * pl, pl * sizeof(float));
        for (int i = 0; i < gc.cells; i++) h_pk_board[(size_t)n * gc.cells + i] = h_lboard[(size_t)q * CELLS_PAD + i];
        h_pk_tomove[n] = h_player[q]; h_pk_mn[n] = h_ply[q]; h_pk_game[n] = g;
        qs.push_back(q); ags.push_back(agent);
        n++;
      }
    }
    if (n == 0) continue;
    agz_leaf_batch b{};
    b.n = n; b.features = gc.F; b.height = gc.m; b.width = gc.n; b.policy_len = plen;
    b.planes = h_pk_planes; b.board = h_pk_board; b.to_move = h_pk_tomove; b.move_number = h_pk_mn; b.game = h_pk_game;
    b.policy = h_pk_policy; b.value = h_pk_value;
    memset(h_pk_policy, 0, (size_t)n * plen * sizeof(float));
    memset(h_pk_value, 0, (size_t)n * sizeof(float));
    const int rc = cb_fn[a](cb_user[a], &b);
    cb_calls++; cb_leaves += n;
    if (rc != 0) { agz::set_error("the host inferencer (AGZ_INF_CALLBACK, agent %d) returned %d for a batch of %d leaves: the search is aborted, reset the arena", a, rc, n); return AGZ_E_CALLBACK; }
    for (int i = 0; i < n; i++) {
      const int q = qs[i], g = q / d.V, l = q - g * d.V, ag = ags[i];
      const size_t slot = (size_t)l * G + slot_host[g];     // lane-major NN slots, as k_select / k_expand index them
      memcpy(h_policy[ag] + slot * plen, h_pk_policy + (size_t)i * plen, (size_t)plen * sizeof(float));
      h_value[ag][slot] = h_pk_value[i];
    }
  }
  for (int a = 0; a < 2; a++) {
    if (inf_kind[a] != AGZ_INF_CALLBACK) continue;
    AGZ_HIP_TRY(hipMemcpyAsync(d_policy[a], h_policy[a], (size_t)rows * cb_plen[a] * sizeof(float), hipMemcpyHostToDevice, s));
    AGZ_HIP_TRY(hipMemcpyAsync(d_value[a], h_value[a], (size_t)rows * sizeof(float), hipMemcpyHostToDevice, s));
  }
  return AGZ_OK;
}

// NN batch slots.  One shared net (or synthetic inferencers): slot = game.  Two different nets: games whose
// current agent is A take slots [0, nA), B's take [nA, G) so that each net runs one dense sub-batch.
int agz_arena::update_slots() {
  slot_host.resize(G);
  if (!split_nets()) {
    for (int g = 0; g < G; g++) slot_host[g] = g;
    nA_slots = G; nB_slots = G;
  } else {
    // who moves in game g is read from the device state (the Arena's currentPlayer: A moves when its colour is to move),
    // so games may sit at different plies (agz_arena_apply_moves with AGZ_NO_MOVE / rejected moves, random openings)
    std::vector<int32_t> tm(G), ab(G);
    AGZ_HIP_TRY(hipMemcpyAsync(tm.data(), d.to_move, G * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    AGZ_HIP_TRY(hipMemcpyAsync(ab.data(), d.a_is_black, G * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    int na = 0;
    for (int g = 0; g < G; g++) { bool a_moves = (tm[g] == AGZ_BLACK) == (ab[g] != 0); if (a_moves) na++; }
    int ia = 0, ib = na;
    for (int g = 0; g < G; g++) { bool a_moves = (tm[g] == AGZ_BLACK) == (ab[g] != 0); slot_host[g] = a_moves ? ia++ : ib++; }
    nA_slots = na; nB_slots = G - na;
  }
  AGZ_HIP_TRY(hipMemcpyAsync(d.slot_of_game, slot_host.data(), G * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
  AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
  return AGZ_OK;
}

int agz_arena::nn_step(int prep, int nl) {
  // select -> network -> expand, all asynchronous on the ctx stream
  float* act0 = nullptr; float* act1 = nullptr;
  const size_t slot_elems = (size_t)(gc.m + 2) * (gc.n + 2) * 32;
  if (inf_kind[0] == AGZ_INF_NET) act0 = net[0]->d_act_in;
  if (inf_kind[1] == AGZ_INF_NET) act1 = split_nets() ? net[1]->d_act_in : (net[1]->d_act_in);
  if (split_nets()) { /* both nets see global slot indices; net B's sub-batch starts at slot nA */ }
  const bool split_lanes = d.V > 1;
  {
    ProfScope ps(ctx, AGZ_PROF_SELECT);
    if (split_lanes) {
      hipLaunchKernelGGL(k_select_paths, dim3(G), dim3(64), 0, ctx->stream, d, gc, mc, prep, nl);
      hipLaunchKernelGGL(k_leaf, dim3(G * nl), dim3(64), 0, ctx->stream, d, gc, mc, act0, act1, prep, nl);
    } else {
      hipLaunchKernelGGL(k_select, dim3(G), dim3(64), 0, ctx->stream, d, gc, mc, act0, act1, prep, nl);
    }
  }
  // prepareRoot evaluates the network only for a root without children (search.go:392-408).  With tree reuse that is the rare
  // case: if NO game of this arena needs it, the batched forward is skipped (one 8-byte read-back per move)
  bool need_forward = true;
  int prep_batch = 0;              // > 0: prepareRoot's network batch, packed to the front (k_prep_compact)
  if (prep) {
    unsigned long long cnt = 0;
    AGZ_HIP_TRY(hipMemcpyAsync(&cnt, d.counters + CNT_PREP_EXPAND, 8, hipMemcpyDeviceToHost, ctx->stream));
    AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    need_forward = cnt != prep_expand_seen;
    const unsigned long long n_need = cnt - prep_expand_seen;
    prep_expand_seen = cnt;
    // one shared network, identity slots: evaluate the n_need roots as a batch of their own when a smaller batch runs the same
    // kernels (agz_net::min_same_batch: per board the results are those of the whole-arena batch, bit for bit)
    last_prep_roots = (int)n_need; last_prep_batch = need_forward ? G : 0;
    if (need_forward && prep_compact && !split_nets() && nl == 1 && inf_kind[0] == AGZ_INF_NET && inf_kind[1] == AGZ_INF_NET && n_need < (unsigned long long)G) {
      const int nb = net[0]->min_same_batch((int)n_need, G);
      if (nb < G) {
        const int slot_f4 = (int)(slot_elems / 4);
        hipLaunchKernelGGL(k_prep_compact, dim3(ceil_div(slot_f4, 256)), dim3(256), 0, ctx->stream, d, act0, slot_f4);
        prep_batch = nb; last_prep_batch = nb;
      }
    }
  }
  Dev dx = d;                      // what k_expand reads its network rows through
  if (prep_batch > 0) dx.slot_of_game = d.prep_slot;
  InfDesc inf{};
  for (int a = 0; a < 2; a++) {
    inf.kind[a] = inf_kind[a];
    inf.policy[a] = d_policy[a]; inf.value[a] = d_value[a];
    inf.dummy_player[a] = 0;  // useDummy captures Agent.Player before colours are drawn: None (agogo.go:83-87, agent.go:105-113)
    switch (inf_kind[a]) {
      case AGZ_INF_NET: inf.policy_len[a] = net[a]->conf.ActionSpace; break;
      case AGZ_INF_DUMMY: inf.policy_len[a] = gc.A; break;
      case AGZ_INF_SCRIPT: inf.policy_len[a] = 10; break;
      case AGZ_INF_HASH: inf.policy_len[a] = gc.A + 1; break;
      case AGZ_INF_CALLBACK: inf.kind[a] = AGZ_INF_NET; inf.policy_len[a] = cb_plen[a]; break;   // k_expand reads rows the host function filled
      default: inf.policy_len[a] = 25; break;
    }
  }
  if (!need_forward) {
    if (inf_kind[0] == AGZ_INF_NET && inf_kind[1] == AGZ_INF_NET && !split_nets()) { inf.policy[1] = d_policy[0]; inf.value[1] = d_value[0]; }
  } else if (!split_nets()) {
    agz_net* n = inf_kind[0] == AGZ_INF_NET ? net[0] : (inf_kind[1] == AGZ_INF_NET ? net[1] : nullptr);
    if (n) {
      int a = inf_kind[0] == AGZ_INF_NET ? 0 : 1;
      int r = n->forward_packed(prep_batch > 0 ? prep_batch : G * nl, d_policy[a], d_value[a]);   // lane-major slots: one dense batch
      if (r != AGZ_OK) return r;
      if (inf_kind[0] == AGZ_INF_NET && inf_kind[1] == AGZ_INF_NET) { inf.policy[1] = d_policy[0]; inf.value[1] = d_value[0]; }
    }
  } else {
    // net A on slots [0,nA), net B on slots [nA,G): each net's input buffer is indexed by global slot
    if (nA_slots > 0) { int r = net[0]->forward_packed(nA_slots, d_policy[0], d_value[0]); if (r != AGZ_OK) return r; }
    if (nB_slots > 0) {
      agz_net* nb = net[1];
      float* save = nb->d_act_in;
      nb->d_act_in = save + (size_t)nA_slots * slot_elems;
      int r = nb->forward_packed(nB_slots, d_policy[1] + (size_t)nA_slots * nb->conf.ActionSpace, d_value[1] + nA_slots);
      nb->d_act_in = save;
      if (r != AGZ_OK) return r;
    }
  }
  // host inferencers (AGZ_INF_CALLBACK) AFTER the network pass of the other agent is enqueued: the callee may itself call into a net of
  // this context (its input buffer held this step's leaves until now)
  if (d.cb_mask && need_forward) { int r = cb_run(nl); if (r != AGZ_OK) return r; }
  {
    ProfScope ps(ctx, AGZ_PROF_EXPAND);
    if (split_lanes) {
      hipLaunchKernelGGL(k_expand_prep, dim3(G * nl), dim3(64), 0, ctx->stream, dx, gc, mc, inf, nl);
      hipLaunchKernelGGL(k_expand_commit, dim3(G), dim3(64), 0, ctx->stream, dx, gc, mc, prep, nl);
    } else {
      hipLaunchKernelGGL(k_expand, dim3(G), dim3(64), 0, ctx->stream, dx, gc, mc, inf, prep, nl);
    }
  }
  AGZ_HIP_TRY(hipGetLastError());
  return AGZ_OK;
}

// rows kept[i] of src -> rows i of dst (rowlen floats / words each)
__global__ void k_gather_rows(float* __restrict__ dst, const float* __restrict__ src, const int32_t* __restrict__ idx, int n, int rowlen) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (size_t)n * rowlen) return;
  const int r = (int)(g / rowlen), c = (int)(g - (size_t)r * rowlen);
  dst[(size_t)r * rowlen + c] = src[(size_t)idx[r] * rowlen + c];
}

extern "C" {

int agz_arena_create(agz_ctx* ctx, const agz_game_conf* game, const agz_mcts_conf* mcts, int n_games, uint64_t seed,
                     int max_nodes, agz_arena** out) {
  AGZ_REQUIRE(ctx && game && mcts && out, AGZ_E_INVALID, "agz_arena_create: NULL argument");
  AGZ_REQUIRE(n_games >= 1, AGZ_E_INVALID, "agz_arena_create: n_games must be >= 1");
  AGZ_REQUIRE(mcts->PUCT > 0 && mcts->PUCT <= 1, AGZ_E_INVALID, "MCTSConf is not valid (PUCT must be in (0,1], tree.go:42-44)");
  AGZ_REQUIRE(game->kind >= 0 && game->kind <= 3, AGZ_E_INVALID, "agz_arena_create: unknown game kind %d", game->kind);
  AGZ_REQUIRE(game->m >= 1 && game->n >= 1 && game->m * game->n <= 361, AGZ_E_UNSUPPORTED, "board %dx%d unsupported (max 361 cells)", game->m, game->n);
  if (game->kind == AGZ_GAME_KOMI || game->kind == AGZ_GAME_WQ)
    AGZ_REQUIRE(game->m == game->n, AGZ_E_UNSUPPORTED, "komi/wq boards must be square on device (the reference's non-square geometry is degenerate, game/naughty.go:9-19)");
  AGZ_REQUIRE(game->encoder == AGZ_ENC_TWOPLANE || game->encoder == AGZ_ENC_WQ, AGZ_E_INVALID, "unknown encoder");
  AGZ_REQUIRE(mcts->Budget >= 0, AGZ_E_INVALID, "Budget must be >= 0");
  AGZ_HIP_TRY(hipSetDevice(ctx->device));
  agz_arena* a = new agz_arena();
  a->ctx = ctx; a->G = n_games; a->seed = seed; a->seed0 = seed;
  a->pool_grow = max_nodes <= 0;   // the library chose the pool size: it also keeps it sufficient (AGZ_POOL_GROW); an explicit max_nodes is the caller's budget (AGZ_POOL_STRICT)
  GameCfg& c = a->gc;
  c.kind = game->kind; c.m = game->m; c.n = game->n; c.k = game->k; c.cells = c.m * c.n;
  c.A = c.kind == AGZ_GAME_C4 ? c.n : c.cells;
  c.max_moves = game->max_moves > 0 ? game->max_moves : 2 * c.cells;
  c.encoder = game->encoder; c.F = c.encoder == AGZ_ENC_WQ ? 18 : 2;
  c.komi = game->komi;
  c.flip_in_tree = c.kind != AGZ_GAME_C4;
  c.pass_legal = (c.kind == AGZ_GAME_WQ || c.kind == AGZ_GAME_C4);
  c.go_like = (c.kind == AGZ_GAME_KOMI || c.kind == AGZ_GAME_WQ);
  c.has_passes = c.kind == AGZ_GAME_WQ;
  MctsCfg& m = a->mc;
  m.PUCT = mcts->PUCT; m.maxDepth = mcts->M * mcts->N; m.RandomCount = mcts->RandomCount; m.Budget = mcts->Budget;
  m.RandomMinVisits = mcts->RandomMinVisits; m.RandomTemperature = mcts->RandomTemperature; m.DumbPass = mcts->DumbPass;
  m.ResignPercentage = mcts->ResignPercentage; m.PassPreference = mcts->PassPreference;
  AGZ_REQUIRE(m.maxDepth + 2 <= MAXPATH, AGZ_E_UNSUPPORTED, "M*N too large");
  Dev& d = a->d;
  const int G = n_games, T = 2 * G;
  d.G = G; d.T = T;
  // (a single tree with Budget <= 0 = a search that stops on the wall clock, agz_mcts_set_timeout_ms: no simulation count to size by —
  // room for 65536 expansions, at most the 8 M-node ceiling; a full pool ends such a search, agz_mcts_search)
  // Budget <= 0, explicitly: a SINGLE tree may search by the wall clock (agz_mcts_set_timeout_ms) — no simulation count to size by,
  // room for 65536 expansions unless the caller passes max_nodes (the Go shim derives it from the Timeout).  An arena of SEVERAL games has no
  // wall-clock rule: Budget 0 there means exactly zero simulations per move — every move comes from prepareRoot's one expansion
  // (tests/test_engine_edges_gpu.py) — and two expansions' worth of nodes is all its trees can ever hold.
  const long long size_by = mcts->Budget > 0 ? mcts->Budget : (n_games == 1 ? 65536 : 0);
  // default: FOUR searches' worth of expansions.  A search adds at most Budget + 1 expansions to the subtree kept from the one before; with
  // a kept fraction f per move a tree settles at (Budget + 1)(A + 1) / (1 - f) nodes: two searches' worth (rounds 1-5) is f <= 0.5 — enough for
  // the wide trees of near-uniform priors (the headline keeps < 5 %), NOT for the narrow trees of a peaked policy, which keep most of
  // their nodes move after move (round 6: tests/test_deep_tree_fuzz_gpu.py overflowed it within five moves; one of 512 distinct 9x9 games
  // did in bench.py).  Four is f <= 0.75; beyond that the caller passes max_nodes (the reference is unbounded up to MAXTREESIZE).
  long long cap = max_nodes > 0 ? max_nodes : (long long)(4 * (size_by + 2)) * (c.A + 1) + 16;
  if (cap > 8000000) cap = 8000000;
  d.cap = (int)cap;
  d.moves_stride = c.max_moves + 4;
  int r = AGZ_OK;
#define AL(p, n) if ((r = a->alloc(&d.p, (size_t)(n))) != AGZ_OK) { agz_arena_destroy(a); return r; }
  AL(board, (size_t)G * CELLS_PAD) AL(ring, (size_t)G * RING * CELLS_PAD) AL(to_move, G) AL(ply, G) AL(passes, G)
  AL(pass_count, G) AL(ended, G) AL(winner, G) AL(a_is_black, G) AL(last_move, G) AL(cap_b, G) AL(cap_w, G) AL(zhash, G)
  AL(moves, (size_t)G * d.moves_stride) AL(amoves, (size_t)G * d.moves_stride) AL(n_amoves, G) AL(hist_from, G)
  size_t pool = (size_t)T * 2 * d.cap;
  AL(prior, pool) AL(visits, pool) AL(bsum, pool) AL(kids_off, pool) AL(kids_n, pool) AL(nmove, pool)
  AL(n_nodes, T) AL(cur_pool, T) AL(has_root, T) AL(has_prev, T) AL(prev_ply, T) AL(prev_board, (size_t)T * CELLS_PAD)
  AL(stalled, T) AL(overflow, T) AL(pc_hash, (size_t)T * d.moves_stride) AL(pc_move, (size_t)T * d.moves_stride) AL(pc_n, T) AL(rng, T) AL(rng_game, G)
  d.V = 1;
  AL(vl, pool)
  AL(slot_of_game, G) AL(prep_slot, G) AL(leaf_kind, G) AL(leaf_player, G) AL(leaf_ply, G) AL(leaf_result, G) AL(leaf_board, (size_t)G * CELLS_PAD)
  AL(leaf_legal, (size_t)G * CELLS_PAD) AL(path, (size_t)G * MAXPATH) AL(path_len, G) AL(counters, CNT_N)
  {
    size_t per_ex = (size_t)c.F * c.cells + (c.A + 1) + 3;
    size_t want = std::max<size_t>((size_t)G * c.max_moves * 2, 16384);  // room for restarted games (agz_arena_selfplay)
    // <= 12 GiB of examples per arena (round 5: 2 GiB = 78 k rows on 19x19 — 512 complete games record 370 k, and the complete-games run
    // of round 6 dropped 276 k of them; 288 GB of HBM hold the whole epoch: 512 games x 722 moves x 27.5 KB = 10.2 GB)
    size_t budget = (size_t)(12ull << 30) / (per_ex * 4);
    d.ex_cap = (int)std::min(want, budget);
  }
  AL(ex_planes, (size_t)d.ex_cap * c.F * c.cells) AL(ex_policy, (size_t)d.ex_cap * (c.A + 1)) AL(ex_value, d.ex_cap)
  AL(ex_game, d.ex_cap) AL(ex_prev, d.ex_cap) AL(ex_last, G) AL(ex_count, 1) AL(ex_labelled, d.ex_cap) AL(best_out, G)
  {  // zobrist keys: komi draws only the first size+1 table entries (komi/zobrist.go:38); wq draws all (wq/zobrist.go:31-42)
    std::vector<int32_t> zt((size_t)2 * c.cells, 0);
    SplitMix64 rr(1337);
    int lim = c.kind == AGZ_GAME_KOMI ? std::min(c.cells + 1, 2 * c.cells) : 2 * c.cells;
    for (int i = 0; i < lim; i++) zt[i] = rr.int31();
    int32_t* zp = nullptr;
    if ((r = a->alloc(&zp, zt.size())) != AGZ_OK) { agz_arena_destroy(a); return r; }
    if (hipMemcpyAsync(zp, zt.data(), zt.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
      agz_arena_destroy(a);
      agz::set_error("arena: zobrist table upload failed");
      return AGZ_E_HIP;
    }
    d.ztable = zp;
  }
#undef AL
  a->a_is_black.assign(G, 1);
  *out = a;
  return agz_arena_reset(a, nullptr);
}

void agz_arena_destroy(agz_arena* a) {
  if (!a) return;
  hipSetDevice(a->ctx->device);
  hipStreamSynchronize(a->ctx->stream);
  for (void* p : a->allocs) hipFree(p);
  for (void* p : a->host_allocs) hipHostFree(p);
  for (int i = 0; i < 2; i++) { if (a->d_policy[i]) hipFree(a->d_policy[i]); if (a->d_value[i]) hipFree(a->d_value[i]); }
  if (a->d_forced) hipFree(a->d_forced);
  if (a->d_remaining) hipFree(a->d_remaining);
  delete a;
}

int agz_arena_set_inferencer(agz_arena* a, int agent, int kind, agz_net* net) {
  AGZ_REQUIRE(a && (agent == 0 || agent == 1), AGZ_E_INVALID, "agz_arena_set_inferencer: bad agent");
  AGZ_REQUIRE(kind >= AGZ_INF_NET && kind <= AGZ_INF_UNIFORM, AGZ_E_INVALID, "unknown inferencer kind %d%s", kind, kind == AGZ_INF_CALLBACK ? " (AGZ_INF_CALLBACK is set by agz_arena_set_inferencer_callback)" : "");
  if (kind == AGZ_INF_NET) {
    AGZ_REQUIRE(net && net->committed, AGZ_E_STATE, "AGZ_INF_NET needs a committed net");
    AGZ_REQUIRE(net->ctx == a->ctx, AGZ_E_INVALID, "net belongs to another ctx");
    AGZ_REQUIRE(net->conf.Height == a->gc.m && net->conf.Width == a->gc.n && net->conf.Features == a->gc.F, AGZ_E_INVALID,
                "net geometry (%dx%dx%d) does not match the game/encoder (%dx%dx%d)", net->conf.Height, net->conf.Width,
                net->conf.Features, a->gc.m, a->gc.n, a->gc.F);
    AGZ_REQUIRE(net->conf.ActionSpace >= a->gc.A, AGZ_E_INVALID, "net ActionSpace %d < game ActionSpace %d", net->conf.ActionSpace, a->gc.A);
    AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
    int r = net->ensure_batch(a->G * a->d.V);
    if (r != AGZ_OK) return r;
    if (a->d_policy[agent]) { hipFree(a->d_policy[agent]); a->d_policy[agent] = nullptr; }
    if (a->d_value[agent]) { hipFree(a->d_value[agent]); a->d_value[agent] = nullptr; }
    AGZ_HIP_TRY(hipMalloc(&a->d_policy[agent], (size_t)a->G * a->d.V * net->conf.ActionSpace * sizeof(float)));
    AGZ_HIP_TRY(hipMalloc(&a->d_value[agent], (size_t)a->G * a->d.V * sizeof(float)));
  }
  {  // two different nets evaluate an A and a B sub-batch of ONE lane per game (nn_step): refuse the combination with lanes
    const int other = agent ^ 1;
    const bool would_split = kind == AGZ_INF_NET && a->inf_kind[other] == AGZ_INF_NET && a->net[other] != net;
    AGZ_REQUIRE(!(would_split && a->d.V > 1), AGZ_E_UNSUPPORTED,
                "agz_arena_set_inferencer: agents with two different nets search one lane at a time (agz_arena_set_parallel(1) first)");
  }
  a->inf_kind[agent] = kind;
  a->net[agent] = kind == AGZ_INF_NET ? net : nullptr;
  a->cb_fn[agent] = nullptr; a->cb_user[agent] = nullptr; a->cb_plen[agent] = 0;
  a->d.cb_mask &= ~(1 << agent);
  return a->update_slots();
}

int agz_arena_set_inferencer_callback(agz_arena* a, int agent, agz_infer_fn fn, void* user, int policy_len) {
  AGZ_REQUIRE(a && (agent == 0 || agent == 1), AGZ_E_INVALID, "agz_arena_set_inferencer_callback: bad agent");
  AGZ_REQUIRE(fn, AGZ_E_INVALID, "agz_arena_set_inferencer_callback: NULL function");
  AGZ_REQUIRE(!a->in_move, AGZ_E_STATE, "agz_arena_set_inferencer_callback: a search is in progress");
  AGZ_REQUIRE(policy_len >= a->gc.A && policy_len <= 4096, AGZ_E_INVALID,
              "agz_arena_set_inferencer_callback: policy_len %d (the game's ActionSpace is %d; the pass probability is the LAST entry, search.go:276)", policy_len, a->gc.A);
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  AGZ_HIP_TRY(hipStreamSynchronize(a->ctx->stream));
  a->inf_kind[agent] = AGZ_INF_CALLBACK;
  a->net[agent] = nullptr;
  a->cb_fn[agent] = fn; a->cb_user[agent] = user; a->cb_plen[agent] = policy_len;
  a->d.cb_mask |= 1 << agent;
  int r = a->cb_setup();
  if (r != AGZ_OK) return r;
  return a->update_slots();
}

int agz_arena_set_pool_policy(agz_arena* a, int policy) {
  AGZ_REQUIRE(a, AGZ_E_INVALID, "arena is NULL");
  AGZ_REQUIRE(policy == AGZ_POOL_STRICT || policy == AGZ_POOL_STOP_SEARCH || policy == AGZ_POOL_GROW, AGZ_E_INVALID, "agz_arena_set_pool_policy: policy %d", policy);
  AGZ_REQUIRE(!a->in_move, AGZ_E_STATE, "agz_arena_set_pool_policy: a search is in progress");
  a->pool_stop = policy == AGZ_POOL_STOP_SEARCH;
  a->pool_grow = policy == AGZ_POOL_GROW;
  return AGZ_OK;
}

int agz_arena_set_parallel(agz_arena* a, int lanes) {
  AGZ_REQUIRE(a, AGZ_E_INVALID, "arena is NULL");
  AGZ_REQUIRE(lanes >= 1 && lanes <= 16, AGZ_E_INVALID, "agz_arena_set_parallel: lanes must be in [1,16], got %d", lanes);
  AGZ_REQUIRE(!a->in_move, AGZ_E_STATE, "agz_arena_set_parallel: a search is in progress");
  AGZ_REQUIRE(lanes == 1 || !a->split_nets(), AGZ_E_UNSUPPORTED, "agz_arena_set_parallel: agents with two different nets search one lane at a time");
  if (lanes == a->d.V) return AGZ_OK;
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  AGZ_HIP_TRY(hipStreamSynchronize(a->ctx->stream));
  Dev& d = a->d;
  const size_t n = (size_t)a->G * lanes;
  int r = AGZ_OK;
  // the previous (smaller) scratch arrays stay on the arena's allocation list and are released with it
#define RL(p, cnt) if ((r = a->alloc(&d.p, (size_t)(cnt))) != AGZ_OK) return r;
  RL(leaf_kind, n) RL(leaf_player, n) RL(leaf_ply, n) RL(leaf_result, n) RL(leaf_board, n * CELLS_PAD) RL(leaf_legal, n * CELLS_PAD)
  RL(path, n * MAXPATH) RL(path_len, n)
  RL(exp_score, n * CELLS_PAD) RL(exp_move, n * CELLS_PAD) RL(exp_n, n) RL(exp_value, n)
#undef RL
  d.V = lanes;
  AGZ_HIP_TRY(hipStreamSynchronize(a->ctx->stream));
  for (int agent = 0; agent < 2; agent++)   // network batch and output buffers grow to G*lanes rows
    if (a->inf_kind[agent] == AGZ_INF_NET) { r = agz_arena_set_inferencer(a, agent, AGZ_INF_NET, a->net[agent]); if (r != AGZ_OK) return r; }
  if (a->d.cb_mask) { r = a->cb_setup(); if (r != AGZ_OK) return r; }   // host-callback agents: staging for G * lanes leaves
  return AGZ_OK;
}

int agz_arena_reset(agz_arena* a, const uint8_t* a_is_black) {
  AGZ_REQUIRE(a, AGZ_E_INVALID, "arena is NULL");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  hipStream_t s = a->ctx->stream;
  uint8_t* dab = nullptr;
  if (a_is_black) {
    AGZ_HIP_TRY(hipMalloc(&dab, a->G));
    AGZ_HIP_TRY(hipMemcpyAsync(dab, a_is_black, a->G, hipMemcpyHostToDevice, s));
  }
  AGZ_HIP_TRY(hipMemsetAsync(a->d.counters, 0, CNT_N * sizeof(unsigned long long), s));
  AGZ_HIP_TRY(hipMemsetAsync(a->d.ex_count, 0, sizeof(int32_t), s));
  hipLaunchKernelGGL(k_reset, dim3(a->G), dim3(64), 0, s, a->d, a->gc, dab, (unsigned long long)a->seed, (unsigned long long)a->seed0);
  std::vector<int32_t> ab(a->G);
  AGZ_HIP_TRY(hipMemcpyAsync(ab.data(), a->d.a_is_black, a->G * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipStreamSynchronize(s));
  if (dab) hipFree(dab);
  for (int g = 0; g < a->G; g++) a->a_is_black[g] = (uint8_t)ab[g];
  a->in_move = false; a->moves_done = 0; a->prep_expand_seen = 0;
  a->seed += 0x9E3779B97F4A7C15ull;  // next reset draws new colours
  return a->update_slots();
}

int agz_arena_begin_move(agz_arena* a) {
  AGZ_REQUIRE(a, AGZ_E_INVALID, "arena is NULL");
  AGZ_REQUIRE(!a->in_move, AGZ_E_STATE, "agz_arena_begin_move: previous move not ended");
  for (int i = 0; i < 2; i++) AGZ_REQUIRE(a->inf_kind[i] != AGZ_INF_NET || a->net[i], AGZ_E_STATE, "agent %d has no net", i);
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  if (a->split_nets()) { int r = a->update_slots(); if (r != AGZ_OK) return r; }
  {
    ProfScope ps(a->ctx, AGZ_PROF_MOVE);
    hipLaunchKernelGGL(k_begin_move, dim3(a->G), dim3(64), 0, a->ctx->stream, a->d, a->gc, a->mc);
  }
  a->in_move = true;
  a->sims_this_move = 0; a->room_sims = 0;
  if (a->pool_grow) {   // room for this move's search before its first expansion (prepareRoot's): AGZ_POOL_GROW
    const long long sims = a->mc.Budget > 0 ? a->mc.Budget : 0;
    int r = a->ensure_pool_room(sims);
    if (r != AGZ_OK) { a->in_move = false; return r; }
    a->room_sims = sims;
  }
  return a->nn_step(1);  // prepareRoot
}

int agz_arena_debug_counter(agz_arena* a, int which, int64_t* value) {
  AGZ_REQUIRE(a && value && which >= 0 && which < CNT_N, AGZ_E_INVALID, "agz_arena_debug_counter: bad argument");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  unsigned long long v = 0;
  AGZ_HIP_TRY(hipMemcpyAsync(&v, a->d.counters + which, 8, hipMemcpyDeviceToHost, a->ctx->stream));
  AGZ_HIP_TRY(hipStreamSynchronize(a->ctx->stream));
  *value = (int64_t)v;
  return AGZ_OK;
}

int agz_arena_pool_capacity(agz_arena* a, int* nodes_per_pool, int* grows) {
  AGZ_REQUIRE(a && nodes_per_pool && grows, AGZ_E_INVALID, "agz_arena_pool_capacity: NULL argument");
  *nodes_per_pool = a->d.cap; *grows = a->pool_grows;
  return AGZ_OK;
}

int agz_arena_set_prep_compact(agz_arena* a, int on) {
  AGZ_REQUIRE(a, AGZ_E_INVALID, "arena is NULL");
  a->prep_compact = on != 0;
  return AGZ_OK;
}

int agz_arena_last_prep_batch(agz_arena* a, int* boards, int* roots) {
  AGZ_REQUIRE(a && boards && roots, AGZ_E_INVALID, "agz_arena_last_prep_batch: NULL argument");
  *boards = a->last_prep_batch; *roots = a->last_prep_roots;
  return AGZ_OK;
}

int agz_arena_simulate(agz_arena* a, int k) {
  AGZ_REQUIRE(a, AGZ_E_INVALID, "arena is NULL");
  AGZ_REQUIRE(a->in_move, AGZ_E_STATE, "agz_arena_simulate: call agz_arena_begin_move first");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  if (a->pool_grow && a->sims_this_move + k > a->room_sims) {   // more simulations than the move's Budget promised (or a wall-clock search): room for these
    int r = a->ensure_pool_room(k);
    if (r != AGZ_OK) return r;
    a->room_sims = a->sims_this_move + k;
  }
  a->sims_this_move += k;
  // k simulations per tree, in rounds of up to V lanes (agz_arena_set_parallel; V = 1: one at a time)
  for (int done = 0; done < k;) {
    int nl = std::min(a->d.V, k - done);
    int r = a->nn_step(0, nl);
    if (r != AGZ_OK) return r;
    done += nl;
  }
  return AGZ_OK;
}

int agz_arena_end_move(agz_arena* a, int record) {
  AGZ_REQUIRE(a, AGZ_E_INVALID, "arena is NULL");
  AGZ_REQUIRE(a->in_move, AGZ_E_STATE, "agz_arena_end_move: call agz_arena_begin_move first");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  {
    ProfScope ps(a->ctx, AGZ_PROF_MOVE);
    hipLaunchKernelGGL(k_end_move, dim3(a->G), dim3(64), 0, a->ctx->stream, a->d, a->gc, a->mc, record, a->restart ? 1 : 0, (const int32_t*)nullptr, 0);
  }
  AGZ_HIP_TRY(hipGetLastError());
  a->in_move = false;
  a->moves_done++;
  return AGZ_OK;
}

int agz_arena_apply_moves(agz_arena* a, const int32_t* moves) {
  AGZ_REQUIRE(a && moves, AGZ_E_INVALID, "agz_arena_apply_moves: NULL argument");
  AGZ_REQUIRE(!a->in_move, AGZ_E_STATE, "agz_arena_apply_moves: a search is in progress (call agz_arena_end_move first)");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  hipStream_t s = a->ctx->stream;
  if (!a->d_forced) AGZ_HIP_TRY(hipMalloc(&a->d_forced, (size_t)a->G * sizeof(int32_t)));
  unsigned long long before = 0, after = 0;
  AGZ_HIP_TRY(hipMemcpyAsync(&before, a->d.counters + CNT_ILLEGAL, 8, hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipMemcpyAsync(a->d_forced, moves, (size_t)a->G * sizeof(int32_t), hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipStreamSynchronize(s));   // `moves` is the caller's buffer
  hipLaunchKernelGGL(k_end_move, dim3(a->G), dim3(64), 0, s, a->d, a->gc, a->mc, 0, 0, (const int32_t*)a->d_forced, 0);
  AGZ_HIP_TRY(hipGetLastError());
  AGZ_HIP_TRY(hipMemcpyAsync(&after, a->d.counters + CNT_ILLEGAL, 8, hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipStreamSynchronize(s));
  a->moves_done++;
  AGZ_REQUIRE(after == before, AGZ_E_INVALID, "agz_arena_apply_moves: %llu illegal move(s) (State.Check failed); those games were left unchanged",
              after - before);
  return AGZ_OK;
}

int agz_arena_get_stats(agz_arena* a, agz_arena_stats* out) {
  AGZ_REQUIRE(a && out, AGZ_E_INVALID, "NULL argument");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  unsigned long long c[CNT_N];
  std::vector<int32_t> ended(a->G);
  AGZ_HIP_TRY(hipMemcpyAsync(c, a->d.counters, sizeof(c), hipMemcpyDeviceToHost, a->ctx->stream));
  AGZ_HIP_TRY(hipMemcpyAsync(ended.data(), a->d.ended, a->G * sizeof(int32_t), hipMemcpyDeviceToHost, a->ctx->stream));
  AGZ_HIP_TRY(hipStreamSynchronize(a->ctx->stream));
  out->sims_total = (int64_t)c[CNT_SIMS]; out->sims_nonnull = (int64_t)c[CNT_NONNULL]; out->nn_evals = (int64_t)c[CNT_EVALS];
  out->moves_played = (int64_t)c[CNT_MOVES]; out->games_finished = (int64_t)c[CNT_GAMES]; out->examples = (int64_t)c[CNT_EXAMPLES];
  out->n_games = a->G; out->tree_full = (int32_t)c[CNT_FULL]; out->examples_dropped = (int32_t)c[CNT_DROPPED];
  out->path_nodes = (int64_t)c[CNT_PATH]; out->children_read = (int64_t)c[CNT_KIDS];
  int act = 0;
  for (int g = 0; g < a->G; g++) act += ended[g] ? 0 : 1;
  out->n_active = act;
  return AGZ_OK;
}

int agz_arena_get_results(agz_arena* a, int64_t* a_wins, int64_t* b_wins, int64_t* draws) {
  AGZ_REQUIRE(a, AGZ_E_INVALID, "arena is NULL");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  unsigned long long c[CNT_N];
  AGZ_HIP_TRY(hipMemcpyAsync(c, a->d.counters, sizeof(c), hipMemcpyDeviceToHost, a->ctx->stream));
  AGZ_HIP_TRY(hipStreamSynchronize(a->ctx->stream));
  if (a_wins) *a_wins = (int64_t)c[CNT_A_WINS];
  if (b_wins) *b_wins = (int64_t)c[CNT_B_WINS];
  if (draws) *draws = (int64_t)c[CNT_DRAWS];
  return AGZ_OK;
}

int agz_arena_play(agz_arena* a, int n_moves, int record) {
  AGZ_REQUIRE(a, AGZ_E_INVALID, "arena is NULL");
  int played = 0;
  while (n_moves <= 0 || played < n_moves) {
    if ((played % 4) == 0 || n_moves <= 0) {  // poll the device for "all games ended"
      agz_arena_stats st;
      int r = agz_arena_get_stats(a, &st);
      if (r != AGZ_OK) return r;
      if (st.tree_full && !a->pool_stop) { agz::set_error("agz_arena_play: %d tree pool(s) overflowed (max_nodes too small)", st.tree_full); return AGZ_E_TREE_FULL; }
      if (st.n_active == 0) break;
    }
    int r = agz_arena_begin_move(a);
    if (r != AGZ_OK) return r;
    r = agz_arena_simulate(a, a->mc.Budget);
    if (r != AGZ_OK) return r;
    r = agz_arena_end_move(a, record);
    if (r != AGZ_OK) return r;
    played++;
  }
  AGZ_HIP_TRY(hipStreamSynchronize(a->ctx->stream));
  return AGZ_OK;
}

int agz_arena_selfplay(agz_arena* a, int64_t n_games_target, int record) {
  AGZ_REQUIRE(a && n_games_target >= 1, AGZ_E_INVALID, "agz_arena_selfplay: bad argument");
  AGZ_REQUIRE(!a->split_nets(), AGZ_E_UNSUPPORTED, "agz_arena_selfplay: agents with two different nets must use agz_arena_play (lockstep plies)");
  AGZ_REQUIRE(!a->in_move, AGZ_E_STATE, "agz_arena_selfplay: a move is in progress");
  a->restart = true;
  int rc = AGZ_OK;
  for (int64_t moves = 0;; moves++) {
    if ((moves & 3) == 0) {
      agz_arena_stats st;
      if ((rc = agz_arena_get_stats(a, &st)) != AGZ_OK) break;
      if (st.tree_full && !a->pool_stop) { agz::set_error("agz_arena_selfplay: %d tree pool(s) overflowed", st.tree_full); rc = AGZ_E_TREE_FULL; break; }
      if (st.games_finished >= n_games_target) break;
    }
    if ((rc = agz_arena_begin_move(a)) != AGZ_OK) break;
    if ((rc = agz_arena_simulate(a, a->mc.Budget)) != AGZ_OK) break;
    if ((rc = agz_arena_end_move(a, record)) != AGZ_OK) break;
  }
  a->restart = false;
  if (rc == AGZ_OK) AGZ_HIP_TRY(hipStreamSynchronize(a->ctx->stream));
  return rc;
}

int agz_arena_get_game(agz_arena* a, int g, int32_t* board, agz_game_state* st) {
  AGZ_REQUIRE(a && g >= 0 && g < a->G, AGZ_E_INVALID, "bad game index");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  hipStream_t s = a->ctx->stream;
  AGZ_HIP_TRY(hipStreamSynchronize(s));
  if (board) {
    std::vector<int8_t> b(CELLS_PAD);
    AGZ_HIP_TRY(hipMemcpy(b.data(), a->d.board + (size_t)g * CELLS_PAD, CELLS_PAD, hipMemcpyDeviceToHost));
    for (int i = 0; i < a->gc.cells; i++) board[i] = b[i];
  }
  if (st) {
    int32_t v[7]; float f[2];
    int32_t* src[7] = {a->d.to_move, a->d.ply, a->d.passes, a->d.ended, a->d.winner, a->d.a_is_black, a->d.last_move};
    for (int i = 0; i < 7; i++) AGZ_HIP_TRY(hipMemcpy(&v[i], src[i] + g, sizeof(int32_t), hipMemcpyDeviceToHost));
    AGZ_HIP_TRY(hipMemcpy(&f[0], a->d.cap_b + g, sizeof(float), hipMemcpyDeviceToHost));
    AGZ_HIP_TRY(hipMemcpy(&f[1], a->d.cap_w + g, sizeof(float), hipMemcpyDeviceToHost));
    st->to_move = v[0]; st->move_number = a->gc.kind == AGZ_GAME_C4 ? 1 : v[1]; st->passes = v[2]; st->ended = v[3]; st->winner = v[4];
    st->a_is_black = v[5]; st->last_move = v[6]; st->reserved = v[1]; st->score_black = f[0]; st->score_white = f[1];
  }
  return AGZ_OK;
}

int agz_arena_get_history(agz_arena* a, int g, int32_t* moves, int cap, int* n) {
  AGZ_REQUIRE(a && g >= 0 && g < a->G && n, AGZ_E_INVALID, "bad argument");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  AGZ_HIP_TRY(hipStreamSynchronize(a->ctx->stream));
  int32_t ply = 0;   // length of the arena's move list (every Search result, arena.go:125), not the game's MoveNumber
  AGZ_HIP_TRY(hipMemcpy(&ply, a->d.n_amoves + g, sizeof(int32_t), hipMemcpyDeviceToHost));
  std::vector<int16_t> mv(a->d.moves_stride);
  AGZ_HIP_TRY(hipMemcpy(mv.data(), a->d.amoves + (size_t)g * a->d.moves_stride, mv.size() * sizeof(int16_t), hipMemcpyDeviceToHost));
  *n = ply;
  for (int i = 0; i < ply && i < cap; i++) moves[i] = mv[i];
  return AGZ_OK;
}

int agz_arena_root_children(agz_arena* a, int g, int agent, int32_t* moves, uint32_t* visits, float* bs, float* priors,
                            int cap, int* n) {
  AGZ_REQUIRE(a && g >= 0 && g < a->G && (agent == 0 || agent == 1) && n, AGZ_E_INVALID, "bad argument");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  AGZ_HIP_TRY(hipStreamSynchronize(a->ctx->stream));
  int t = agent * a->G + g;
  int32_t pool = 0, off = -1, hr = 0;
  int16_t kn = 0;
  AGZ_HIP_TRY(hipMemcpy(&hr, a->d.has_root + t, 4, hipMemcpyDeviceToHost));
  *n = 0;
  if (!hr) return AGZ_OK;
  AGZ_HIP_TRY(hipMemcpy(&pool, a->d.cur_pool + t, 4, hipMemcpyDeviceToHost));
  size_t base = ((size_t)t * 2 + pool) * a->d.cap;
  AGZ_HIP_TRY(hipMemcpy(&off, a->d.kids_off + base, 4, hipMemcpyDeviceToHost));
  AGZ_HIP_TRY(hipMemcpy(&kn, a->d.kids_n + base, 2, hipMemcpyDeviceToHost));
  if (off < 0) return AGZ_OK;
  int k = std::min<int>(kn, cap);
  std::vector<int16_t> mv(k);
  AGZ_HIP_TRY(hipMemcpy(mv.data(), a->d.nmove + base + off, k * sizeof(int16_t), hipMemcpyDeviceToHost));
  for (int i = 0; i < k; i++) moves[i] = mv[i];
  AGZ_HIP_TRY(hipMemcpy(visits, a->d.visits + base + off, k * sizeof(uint32_t), hipMemcpyDeviceToHost));
  AGZ_HIP_TRY(hipMemcpy(bs, a->d.bsum + base + off, k * sizeof(float), hipMemcpyDeviceToHost));
  AGZ_HIP_TRY(hipMemcpy(priors, a->d.prior + base + off, k * sizeof(float), hipMemcpyDeviceToHost));
  *n = kn;
  return AGZ_OK;
}

int agz_arena_tree_nodes(agz_arena* a, int g, int agent, int* n_nodes) {
  AGZ_REQUIRE(a && g >= 0 && g < a->G && (agent == 0 || agent == 1) && n_nodes, AGZ_E_INVALID, "bad argument");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  AGZ_HIP_TRY(hipStreamSynchronize(a->ctx->stream));
  AGZ_HIP_TRY(hipMemcpy(n_nodes, a->d.n_nodes + agent * a->G + g, 4, hipMemcpyDeviceToHost));
  return AGZ_OK;
}

int agz_arena_get_examples(agz_arena* a, float* planes, float* policy, float* value, int32_t* game_idx, int cap, int* n) {
  AGZ_REQUIRE(a && n, AGZ_E_INVALID, "bad argument");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  AGZ_HIP_TRY(hipStreamSynchronize(a->ctx->stream));
  int32_t cnt = 0;
  AGZ_HIP_TRY(hipMemcpy(&cnt, a->d.ex_count, 4, hipMemcpyDeviceToHost));
  cnt = std::min(cnt, a->d.ex_cap);
  *n = cnt;
  int k = std::min(cnt, cap);
  if (k <= 0) return AGZ_OK;
  const GameCfg& c = a->gc;
  if (planes) AGZ_HIP_TRY(hipMemcpy(planes, a->d.ex_planes, (size_t)k * c.F * c.cells * 4, hipMemcpyDeviceToHost));
  if (policy) AGZ_HIP_TRY(hipMemcpy(policy, a->d.ex_policy, (size_t)k * (c.A + 1) * 4, hipMemcpyDeviceToHost));
  if (value) AGZ_HIP_TRY(hipMemcpy(value, a->d.ex_value, (size_t)k * 4, hipMemcpyDeviceToHost));
  if (game_idx) AGZ_HIP_TRY(hipMemcpy(game_idx, a->d.ex_game, (size_t)k * 4, hipMemcpyDeviceToHost));
  return AGZ_OK;
}

int agz_arena_clear_examples(agz_arena* a) {
  AGZ_REQUIRE(a, AGZ_E_INVALID, "arena is NULL");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  AGZ_HIP_TRY(hipMemsetAsync(a->d.ex_count, 0, 4, a->ctx->stream));
  AGZ_HIP_TRY(hipMemsetAsync(a->d.ex_last, 0xff, a->G * 4, a->ctx->stream));
  return AGZ_OK;
}

int agz_arena_drop_labelled_examples(agz_arena* a) {
  AGZ_REQUIRE(a, AGZ_E_INVALID, "arena is NULL");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  hipStream_t s = a->ctx->stream;
  AGZ_HIP_TRY(hipStreamSynchronize(s));
  const auto& c = a->gc;
  int32_t cnt = 0;
  AGZ_HIP_TRY(hipMemcpy(&cnt, a->d.ex_count, 4, hipMemcpyDeviceToHost));
  cnt = std::min(cnt, a->d.ex_cap);
  if (cnt == 0) return AGZ_OK;
  std::vector<uint8_t> lab(cnt);
  std::vector<int32_t> game(cnt), prev(cnt);
  AGZ_HIP_TRY(hipMemcpy(lab.data(), a->d.ex_labelled, (size_t)cnt, hipMemcpyDeviceToHost));
  AGZ_HIP_TRY(hipMemcpy(game.data(), a->d.ex_game, (size_t)cnt * 4, hipMemcpyDeviceToHost));
  AGZ_HIP_TRY(hipMemcpy(prev.data(), a->d.ex_prev, (size_t)cnt * 4, hipMemcpyDeviceToHost));
  std::vector<int32_t> kept, map(cnt, -1);
  for (int i = 0; i < cnt; i++) if (!lab[i]) { map[i] = (int32_t)kept.size(); kept.push_back(i); }
  const int k = (int)kept.size();
  if (k == cnt) return AGZ_OK;                       // nothing labelled
  std::vector<int32_t> last(a->G, -1);
  if (k > 0) {
    // the rows of games still in flight move to the front (through a scratch copy: source and target ranges overlap), their
    // per-game chains (ex_prev, ex_last) are re-linked to the new row numbers
    std::vector<int32_t> ngame(k), nprev(k);
    for (int i = 0; i < k; i++) {
      ngame[i] = game[kept[i]];
      nprev[i] = prev[kept[i]] >= 0 ? map[prev[kept[i]]] : -1;
      last[ngame[i]] = i;                            // rows of a game are recorded in increasing row order
    }
    const size_t xs = (size_t)c.F * c.cells, ps = (size_t)c.A + 1;
    int32_t* d_idx = nullptr;
    float* tmp = nullptr;
    AGZ_HIP_TRY(hipMalloc(&d_idx, (size_t)k * 4));
    hipError_t e = hipMalloc(&tmp, (size_t)k * std::max(xs, ps) * 4);
    if (e != hipSuccess) { hipFree(d_idx); AGZ_HIP_TRY(e); }
    e = hipMemcpyAsync(d_idx, kept.data(), (size_t)k * 4, hipMemcpyHostToDevice, s);
    auto move = [&](float* buf, size_t rowlen) {
      if (e != hipSuccess) return;
      hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)(((size_t)k * rowlen + 255) / 256)), dim3(256), 0, s, tmp, (const float*)buf, (const int32_t*)d_idx, k, (int)rowlen);
      e = hipMemcpyAsync(buf, tmp, (size_t)k * rowlen * 4, hipMemcpyDeviceToDevice, s);
    };
    move(a->d.ex_planes, xs);
    move(a->d.ex_policy, ps);
    move(a->d.ex_value, 1);
    if (e == hipSuccess) e = hipMemcpyAsync(a->d.ex_game, ngame.data(), (size_t)k * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(a->d.ex_prev, nprev.data(), (size_t)k * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemsetAsync(a->d.ex_labelled, 0, (size_t)k, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    hipFree(d_idx); hipFree(tmp);
    AGZ_HIP_TRY(e);
  }
  const int32_t kk = k;
  AGZ_HIP_TRY(hipMemcpyAsync(a->d.ex_last, last.data(), (size_t)a->G * 4, hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipMemcpyAsync(a->d.ex_count, &kk, 4, hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipStreamSynchronize(s));
  return AGZ_OK;
}

int agz_arena_examples_dev(agz_arena* a, float** planes, float** policy, float** value, int* n) {
  AGZ_REQUIRE(a && n, AGZ_E_INVALID, "bad argument");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  AGZ_HIP_TRY(hipStreamSynchronize(a->ctx->stream));
  int32_t cnt = 0;
  AGZ_HIP_TRY(hipMemcpy(&cnt, a->d.ex_count, 4, hipMemcpyDeviceToHost));
  *n = std::min(cnt, a->d.ex_cap);
  if (planes) *planes = a->d.ex_planes;
  if (policy) *policy = a->d.ex_policy;
  if (value) *value = a->d.ex_value;
  return AGZ_OK;
}

int agz_arena_examples_labelled_dev(agz_arena* a, const uint8_t** labelled) {
  AGZ_REQUIRE(a && labelled, AGZ_E_INVALID, "bad argument");
  *labelled = a->d.ex_labelled;
  return AGZ_OK;
}

int agz_arena_random_moves(agz_arena* a, const int32_t* n_moves, uint64_t seed) {
  AGZ_REQUIRE(a && n_moves, AGZ_E_INVALID, "agz_arena_random_moves: NULL argument");
  AGZ_REQUIRE(!a->in_move, AGZ_E_STATE, "agz_arena_random_moves: a search is in progress");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  hipStream_t s = a->ctx->stream;
  if (!a->d_forced) AGZ_HIP_TRY(hipMalloc(&a->d_forced, (size_t)a->G * sizeof(int32_t)));
  if (!a->d_remaining) AGZ_HIP_TRY(hipMalloc(&a->d_remaining, (size_t)a->G * sizeof(int32_t)));
  int most = 0;
  for (int g = 0; g < a->G; g++) {
    AGZ_REQUIRE(n_moves[g] >= 0, AGZ_E_INVALID, "agz_arena_random_moves: n_moves[%d] = %d", g, n_moves[g]);
    most = std::max(most, n_moves[g]);
  }
  AGZ_HIP_TRY(hipMemcpyAsync(a->d_remaining, n_moves, (size_t)a->G * sizeof(int32_t), hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipStreamSynchronize(s));   // `n_moves` is the caller's buffer
  for (int i = 0; i < most; i++) {
    hipLaunchKernelGGL(k_random_pick, dim3(a->G), dim3(64), 0, s, a->d, a->gc, a->d_remaining, (unsigned long long)seed, a->d_forced);
    hipLaunchKernelGGL(k_end_move, dim3(a->G), dim3(64), 0, s, a->d, a->gc, a->mc, 0, 0, (const int32_t*)a->d_forced, 0);
  }
  AGZ_HIP_TRY(hipGetLastError());
  AGZ_HIP_TRY(hipStreamSynchronize(s));
  a->moves_done += most;
  return AGZ_OK;
}

// ---- game.State of one game written from the host: what SetGame hands to a tree (tree.go:120-124) ------------------------
int agz_arena_set_state(agz_arena* a, int g, const agz_state* st) {
  AGZ_REQUIRE(a && st && g >= 0 && g < a->G, AGZ_E_INVALID, "agz_arena_set_state: bad argument");
  AGZ_REQUIRE(!a->in_move, AGZ_E_STATE, "agz_arena_set_state: a search is in progress");
  AGZ_REQUIRE(st->board, AGZ_E_INVALID, "agz_arena_set_state: board is NULL");
  const GameCfg& c = a->gc;
  Dev& d = a->d;
  AGZ_REQUIRE(st->to_move == AGZ_BLACK || st->to_move == AGZ_WHITE, AGZ_E_INVALID, "agz_arena_set_state: to_move %d", st->to_move);
  AGZ_REQUIRE(st->n_moves >= 0 && st->n_moves <= c.max_moves, AGZ_E_INVALID, "agz_arena_set_state: n_moves %d outside [0, max_moves %d]", st->n_moves, c.max_moves);
  AGZ_REQUIRE(st->n_last_moves >= 0 && st->n_last_moves <= st->n_moves && (st->n_last_moves == 0 || st->last_moves), AGZ_E_INVALID, "agz_arena_set_state: bad last_moves");
  AGZ_REQUIRE(st->n_historical >= 0 && st->n_historical <= RING && (st->n_historical == 0 || st->historical), AGZ_E_INVALID, "agz_arena_set_state: n_historical %d outside [0, %d]", st->n_historical, RING);
  AGZ_REQUIRE(st->n_historical <= st->n_moves, AGZ_E_INVALID, "agz_arena_set_state: more historical boards than moves");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  hipStream_t s = a->ctx->stream;
  std::vector<int8_t> b(CELLS_PAD, 0), ring((size_t)RING * CELLS_PAD, 0);
  for (int i = 0; i < c.cells; i++) {
    AGZ_REQUIRE(st->board[i] >= 0 && st->board[i] <= 2, AGZ_E_INVALID, "agz_arena_set_state: board[%d] = %d", i, st->board[i]);
    b[i] = (int8_t)st->board[i];
  }
  const int ply = st->n_moves;
  // ring slot j % RING holds the board after move j (komi's post-move `historical`, DESIGN.md wq): the last n_historical boards
  for (int q = 0; q < st->n_historical; q++) {
    const int j = ply - st->n_historical + 1 + q;   // board after move j, oldest first
    for (int i = 0; i < c.cells; i++) ring[(size_t)(j % RING) * CELLS_PAD + i] = (int8_t)st->historical[(size_t)q * c.cells + i];
  }
  std::vector<int16_t> mv(d.moves_stride, 0);
  for (int q = 0; q < st->n_last_moves; q++) mv[ply - st->n_last_moves + q] = (int16_t)st->last_moves[q];
  const int32_t last = st->n_last_moves > 0 ? st->last_moves[st->n_last_moves - 1] : AGZ_PASS;   // LastMove() of an empty history is Pass
  int32_t pc = 0;   // the arena's consecutive-pass count (arena.go:99-103)
  for (int q = st->n_last_moves - 1; q >= 0 && st->last_moves[q] == AGZ_PASS; q--) pc++;
  const int32_t i32[7] = {st->to_move, ply, st->passes, pc, 0, AGZ_NONE, last};
  int32_t* dst[7] = {d.to_move, d.ply, d.passes, d.pass_count, d.ended, d.winner, d.last_move};
  AGZ_HIP_TRY(hipMemcpyAsync(d.board + (size_t)g * CELLS_PAD, b.data(), CELLS_PAD, hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipMemcpyAsync(d.ring + (size_t)g * RING * CELLS_PAD, ring.data(), ring.size(), hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipMemcpyAsync(d.moves + (size_t)g * d.moves_stride, mv.data(), mv.size() * sizeof(int16_t), hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipMemcpyAsync(d.amoves + (size_t)g * d.moves_stride, mv.data(), mv.size() * sizeof(int16_t), hipMemcpyHostToDevice, s));
  for (int i = 0; i < 7; i++) AGZ_HIP_TRY(hipMemcpyAsync(dst[i] + g, &i32[i], 4, hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipMemcpyAsync(d.n_amoves + g, &ply, 4, hipMemcpyHostToDevice, s));
  const int32_t hist_from = ply - st->n_last_moves;
  AGZ_HIP_TRY(hipMemcpyAsync(d.hist_from + g, &hist_from, 4, hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipMemcpyAsync(d.cap_b + g, &st->captures_black, 4, hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipMemcpyAsync(d.cap_w + g, &st->captures_white, 4, hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipMemcpyAsync(d.zhash + g, &st->hash, 4, hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipStreamSynchronize(s));   // the staging vectors and *st are the caller's / locals
  return AGZ_OK;
}

}  // extern "C"

// ---- mcts.MCTS as a single-tree handle (mcts/tree.go:80-142, mcts/search.go:92-164): a one-game arena whose agent A owns THE tree ----
struct agz_mcts {
  agz_arena* arena = nullptr;
  bool have_game = false;
  // agz_mcts_to_dot: the text of the size query kept for the fill call that follows it (key: what any change of the tree changes)
  std::string dot_cache;
  int64_t dot_key[5] = {-1, -1, -1, -1, -1};
  int timeout_ms = 0;              // agz_mcts_set_timeout_ms: > 0 = the reference's stopping rule (mcts.Config.Timeout), not deterministic
  int64_t last_sims = 0;           // simulations the last Search ran
};

namespace agz {
// Policies (tree.go:128-142) of the game currently set: counts of cachedPolicies[{hash, a}] for a in [0, A], normalised (NaN when none)
__global__ __launch_bounds__(64) void k_policies(Dev d, GameCfg c, int t, float* out) {
  __shared__ int8_t board[CELLS_PAD];
  const int lane = threadIdx.x;
  for (int i = lane; i < c.cells; i += WAVE) board[i] = d.board[i];   // game 0
  __syncthreads();
  const uint32_t hash = c.go_like ? d.zhash[0] : fnv_board(c, board);
  const int pcn = d.pc_n[t];
  float tot = 0.f;
  for (int q = 0; q < pcn; q++) {
    const int mv = d.pc_move[(size_t)t * d.moves_stride + q];
    if (d.pc_hash[(size_t)t * d.moves_stride + q] == hash && mv >= 0 && mv <= c.A) tot += 1.f;
  }
  for (int a = lane; a <= c.A; a += WAVE) {
    float cnt = 0.f;
    for (int q = 0; q < pcn; q++)
      if (d.pc_hash[(size_t)t * d.moves_stride + q] == hash && d.pc_move[(size_t)t * d.moves_stride + q] == a) cnt += 1.f;
    out[a] = __fdiv_rn(cnt, tot);   // 0/0 = NaN, as the reference's retVal[i] /= sum
  }
}
}  // namespace agz

extern "C" {

int agz_mcts_create(agz_ctx* ctx, const agz_game_conf* game, const agz_mcts_conf* conf, uint64_t seed, int max_nodes, agz_mcts** out) {
  AGZ_REQUIRE(out, AGZ_E_INVALID, "agz_mcts_create: NULL argument");
  agz_arena* ar = nullptr;
  int r = agz_arena_create(ctx, game, conf, 1, seed, max_nodes, &ar);
  if (r != AGZ_OK) return r;
  agz_mcts* m = new agz_mcts();
  m->arena = ar;
  *out = m;
  return AGZ_OK;
}

void agz_mcts_destroy(agz_mcts* m) {
  if (!m) return;
  agz_arena_destroy(m->arena);
  delete m;
}

int agz_mcts_set_inferencer(agz_mcts* m, int kind, agz_net* net) {
  AGZ_REQUIRE(m, AGZ_E_INVALID, "mcts is NULL");
  int r = agz_arena_set_inferencer(m->arena, 0, kind, net);
  if (r != AGZ_OK) return r;
  return agz_arena_set_inferencer(m->arena, 1, kind, net);   // agent B's tree is never searched; same inferencer keeps one NN batch
}

int agz_mcts_set_inferencer_callback(agz_mcts* m, agz_infer_fn fn, void* user, int policy_len) {
  AGZ_REQUIRE(m, AGZ_E_INVALID, "mcts is NULL");
  int r = agz_arena_set_inferencer_callback(m->arena, 0, fn, user, policy_len);
  if (r != AGZ_OK) return r;
  return agz_arena_set_inferencer_callback(m->arena, 1, fn, user, policy_len);
}

int agz_mcts_set_pool_policy(agz_mcts* m, int policy) {
  AGZ_REQUIRE(m, AGZ_E_INVALID, "mcts is NULL");
  return agz_arena_set_pool_policy(m->arena, policy);
}

int agz_mcts_set_parallel(agz_mcts* m, int lanes) {
  AGZ_REQUIRE(m, AGZ_E_INVALID, "mcts is NULL");
  return agz_arena_set_parallel(m->arena, lanes);
}

int agz_mcts_set_game(agz_mcts* m, const agz_state* st) {
  AGZ_REQUIRE(m, AGZ_E_INVALID, "mcts is NULL");
  m->dot_cache.clear(); m->dot_cache.shrink_to_fit();   // (a ToDot text kept between its size query and its fill dies with the tree it printed)
  int r = agz_arena_set_state(m->arena, 0, st);
  if (r == AGZ_OK) m->have_game = true;
  return r;
}

int agz_mcts_search(agz_mcts* m, int player, int32_t* best) {
  AGZ_REQUIRE(m && best, AGZ_E_INVALID, "agz_mcts_search: NULL argument");
  AGZ_REQUIRE(player == AGZ_BLACK || player == AGZ_WHITE, AGZ_E_INVALID, "agz_mcts_search: player %d", player);
  agz_arena* a = m->arena;
  m->dot_cache.clear(); m->dot_cache.shrink_to_fit();
  AGZ_REQUIRE(!a->in_move, AGZ_E_STATE, "agz_mcts_search: a search is in progress");
  // (t.current is nil before SetGame in the reference, tree.go:107-111: Search would dereference it)
  AGZ_REQUIRE(m->have_game, AGZ_E_STATE, "agz_mcts_search: no game — call agz_mcts_set_game first (mcts.SetGame, tree.go:107)");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  hipStream_t s = a->ctx->stream;
  // the overflow counter is cumulative over the handle's life: this search is judged by what IT adds
  unsigned long long full0 = 0;
  AGZ_HIP_TRY(hipMemcpyAsync(&full0, a->d.counters + CNT_FULL, 8, hipMemcpyDeviceToHost, s));
  // t.current.SetToMove(player) (search.go:95); the tree of this handle is agent A's: A holds the colour that searches
  const int32_t tm = player, ab = player == AGZ_BLACK ? 1 : 0, zero = 0;
  AGZ_HIP_TRY(hipMemcpyAsync(a->d.to_move, &tm, 4, hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipMemcpyAsync(a->d.a_is_black, &ab, 4, hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipMemcpyAsync(a->d.ended, &zero, 4, hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipStreamSynchronize(s));
  a->a_is_black[0] = (uint8_t)ab;
  int r = agz_arena_begin_move(a);
  if (r != AGZ_OK) return r;
  bool pool_full_stop = false;   // wall-clock search without a Budget, ended by a full node pool: a stopping condition, not an error
  if (m->timeout_ms > 0) {
    // mcts.Config.Timeout (tree.go:18,34; search.go:132-133,196-197): simulations until the wall clock says stop.  Rounds are enqueued
    // in slices and the clock is read after each slice has finished on the device; the slice grows until it takes ~2 ms, so the
    // overshoot stays small next to the reference's 100 ms default.  Budget > 0 still caps the search.
    const auto t0 = std::chrono::steady_clock::now();
    const auto elapsed_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    int64_t done = 0;
    int slice = std::max(1, a->d.V);
    const int64_t cap = a->mc.Budget > 0 ? a->mc.Budget : INT64_MAX;
    while (done < cap && elapsed_ms() < (double)m->timeout_ms) {
      const int n = (int)std::min<int64_t>(slice, cap - done);
      const double before = elapsed_ms();
      r = agz_arena_simulate(a, n);
      // the overflow counter rides on the same synchronisation: a full pool ENDS a wall-clock search (the tree is as large as this
      // handle can hold; the reference's arena is unbounded, its clock is what stops it) instead of spinning on a stalled tree
      unsigned long long full_now = full0;
      if (r == AGZ_OK && (hipMemcpyAsync(&full_now, a->d.counters + CNT_FULL, 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)) {
        agz::set_error("agz_mcts_search: stream synchronisation failed"); r = AGZ_E_HIP;
      }
      if (r != AGZ_OK) { a->in_move = false; return r; }
      done += n;
      if (full_now != full0) { pool_full_stop = a->mc.Budget <= 0; break; }
      if (elapsed_ms() - before < 2.0 && slice < 4096) slice *= 2;
    }
    m->last_sims = done;
  } else {
    r = agz_arena_simulate(a, a->mc.Budget);
    if (r != AGZ_OK) { a->in_move = false; return r; }
    m->last_sims = a->mc.Budget;
  }
  {
    ProfScope ps(a->ctx, AGZ_PROF_MOVE);
    hipLaunchKernelGGL(k_end_move, dim3(1), dim3(64), 0, s, a->d, a->gc, a->mc, 0, 0, (const int32_t*)nullptr, 1);
  }
  a->in_move = false;
  AGZ_HIP_TRY(hipGetLastError());
  int32_t out[2] = {0, 0};
  unsigned long long full = 0;
  AGZ_HIP_TRY(hipMemcpyAsync(&out[0], a->d.best_out, 4, hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipMemcpyAsync(&full, a->d.counters + CNT_FULL, 8, hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipStreamSynchronize(s));
  *best = out[0];
  AGZ_REQUIRE(full == full0 || pool_full_stop || a->pool_stop, AGZ_E_TREE_FULL, "agz_mcts_search: the node pool overflowed (max_nodes too small); the move is the best of the truncated search");
  return AGZ_OK;
}

int agz_mcts_set_timeout_ms(agz_mcts* m, int timeout_ms) {
  AGZ_REQUIRE(m, AGZ_E_INVALID, "mcts is NULL");
  AGZ_REQUIRE(timeout_ms >= 0, AGZ_E_INVALID, "agz_mcts_set_timeout_ms: %d", timeout_ms);
  AGZ_REQUIRE(!m->arena->in_move, AGZ_E_STATE, "agz_mcts_set_timeout_ms: a search is in progress");
  m->timeout_ms = timeout_ms;
  return AGZ_OK;
}

int agz_mcts_last_simulations(agz_mcts* m, int64_t* sims) {
  AGZ_REQUIRE(m && sims, AGZ_E_INVALID, "agz_mcts_last_simulations: NULL argument");
  *sims = m->last_sims;
  return AGZ_OK;
}

int agz_mcts_policies(agz_mcts* m, float* policy, int cap) {
  AGZ_REQUIRE(m && policy, AGZ_E_INVALID, "agz_mcts_policies: NULL argument");
  agz_arena* a = m->arena;
  const int n = a->gc.A + 1;
  AGZ_REQUIRE(cap >= n, AGZ_E_INVALID, "agz_mcts_policies: buffer of %d floats, need %d", cap, n);
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  hipStream_t s = a->ctx->stream;
  float* dp = nullptr;
  AGZ_HIP_TRY(hipMalloc(&dp, (size_t)n * sizeof(float)));
  hipLaunchKernelGGL(k_policies, dim3(1), dim3(64), 0, s, a->d, a->gc, 0, dp);
  hipError_t e = hipMemcpyAsync(policy, dp, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(dp);
  AGZ_HIP_TRY(e);
  return AGZ_OK;
}

int agz_mcts_root_children(agz_mcts* m, int32_t* moves, uint32_t* visits, float* black_scores, float* priors, int cap, int* n) {
  AGZ_REQUIRE(m, AGZ_E_INVALID, "mcts is NULL");
  return agz_arena_root_children(m->arena, 0, 0, moves, visits, black_scores, priors, cap, n);
}

// (*MCTS).Children(of) + the Node fields Log / ToDot print (mcts/unsafe_safe.go:15, graph.go:34, node.go:56-68): the children of
// any node of the live tree — node 0 is the root, child_ids index further calls — so the host can walk or draw the tree.
int agz_mcts_children(agz_mcts* m, int node, int32_t* child_ids, int32_t* moves, uint32_t* visits, float* black_scores, float* priors,
                      int cap, int* n) {
  AGZ_REQUIRE(m && n, AGZ_E_INVALID, "agz_mcts_children: NULL argument");
  agz_arena* a = m->arena;
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  AGZ_HIP_TRY(hipStreamSynchronize(a->ctx->stream));
  int32_t pool = 0, n_nodes = 0, hr = 0, off = -1;
  int16_t kn = 0;
  *n = 0;
  AGZ_HIP_TRY(hipMemcpy(&hr, a->d.has_root, 4, hipMemcpyDeviceToHost));
  if (!hr) return AGZ_OK;
  AGZ_HIP_TRY(hipMemcpy(&pool, a->d.cur_pool, 4, hipMemcpyDeviceToHost));
  AGZ_HIP_TRY(hipMemcpy(&n_nodes, a->d.n_nodes, 4, hipMemcpyDeviceToHost));
  AGZ_REQUIRE(node >= 0 && node < n_nodes, AGZ_E_INVALID, "agz_mcts_children: node %d outside the tree (%d nodes)", node, n_nodes);
  const size_t base = (size_t)pool * a->d.cap;
  AGZ_HIP_TRY(hipMemcpy(&off, a->d.kids_off + base + node, 4, hipMemcpyDeviceToHost));
  AGZ_HIP_TRY(hipMemcpy(&kn, a->d.kids_n + base + node, 2, hipMemcpyDeviceToHost));
  if (off < 0) return AGZ_OK;
  *n = kn;
  const int k = std::min<int>(kn, cap);
  if (k <= 0) return AGZ_OK;
  if (child_ids) for (int i = 0; i < k; i++) child_ids[i] = off + i;
  if (moves) {
    std::vector<int16_t> mv(k);
    AGZ_HIP_TRY(hipMemcpy(mv.data(), a->d.nmove + base + off, k * sizeof(int16_t), hipMemcpyDeviceToHost));
    for (int i = 0; i < k; i++) moves[i] = mv[i];
  }
  if (visits) AGZ_HIP_TRY(hipMemcpy(visits, a->d.visits + base + off, k * sizeof(uint32_t), hipMemcpyDeviceToHost));
  if (black_scores) AGZ_HIP_TRY(hipMemcpy(black_scores, a->d.bsum + base + off, k * sizeof(float), hipMemcpyDeviceToHost));
  if (priors) AGZ_HIP_TRY(hipMemcpy(priors, a->d.prior + base + off, k * sizeof(float), hipMemcpyDeviceToHost));
  return AGZ_OK;
}

// (*MCTS).ToDot (mcts/graph.go:34-90): the live tree as a Graphviz digraph "G", one HTML-table node per tree node (ID, Move,
// Player, Visits, Score, State — the reference's rows; its "Value" row prints the evaluation a node was created with, which the
// device pool does not keep: rendered as "-"), children in move order, the board of a node = its parent's board plus its own move
// (graph.go:57-60,80-82: captures are not applied there either).  The whole pool is copied to the host once.
int agz_mcts_to_dot(agz_mcts* m, int max_nodes, char* buf, size_t cap, size_t* needed) {
  AGZ_REQUIRE(m && needed && (buf || cap == 0), AGZ_E_INVALID, "agz_mcts_to_dot: bad argument");
  agz_arena* a = m->arena;
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  AGZ_HIP_TRY(hipStreamSynchronize(a->ctx->stream));
  int32_t pool = 0, n_nodes = 0, hr = 0;
  AGZ_HIP_TRY(hipMemcpy(&hr, a->d.has_root, 4, hipMemcpyDeviceToHost));
  if (hr) {
    AGZ_HIP_TRY(hipMemcpy(&pool, a->d.cur_pool, 4, hipMemcpyDeviceToHost));
    AGZ_HIP_TRY(hipMemcpy(&n_nodes, a->d.n_nodes, 4, hipMemcpyDeviceToHost));
  }
  if (max_nodes > 0 && n_nodes > max_nodes) n_nodes = max_nodes;       // BFS block order: a prefix of the pool is a top of the tree
  const size_t base = (size_t)pool * a->d.cap;
  {   // the size query builds the text; the fill call that follows reuses it (a 1600-simulation 19x19 tree is hundreds of MB of HTML)
    uint32_t root_vis = 0;
    if (n_nodes) AGZ_HIP_TRY(hipMemcpy(&root_vis, a->d.visits + base, 4, hipMemcpyDeviceToHost));
    const int64_t key[5] = {max_nodes, hr, pool, n_nodes, (int64_t)root_vis};
    if (cap && !m->dot_cache.empty() && std::equal(key, key + 5, m->dot_key)) {
      *needed = m->dot_cache.size() + 1;
      const size_t k = std::min(cap - 1, m->dot_cache.size());
      memcpy(buf, m->dot_cache.data(), k);
      buf[k] = 0;
      m->dot_cache.clear(); m->dot_cache.shrink_to_fit();
      return AGZ_OK;
    }
    std::copy(key, key + 5, m->dot_key);
  }
  std::vector<int32_t> off(n_nodes);
  std::vector<int16_t> kn(n_nodes), mv(n_nodes);
  std::vector<uint32_t> vis(n_nodes);
  std::vector<float> pri(n_nodes);
  if (n_nodes) {
    AGZ_HIP_TRY(hipMemcpy(off.data(), a->d.kids_off + base, (size_t)n_nodes * 4, hipMemcpyDeviceToHost));
    AGZ_HIP_TRY(hipMemcpy(kn.data(), a->d.kids_n + base, (size_t)n_nodes * 2, hipMemcpyDeviceToHost));
    AGZ_HIP_TRY(hipMemcpy(mv.data(), a->d.nmove + base, (size_t)n_nodes * 2, hipMemcpyDeviceToHost));
    AGZ_HIP_TRY(hipMemcpy(vis.data(), a->d.visits + base, (size_t)n_nodes * 4, hipMemcpyDeviceToHost));
    AGZ_HIP_TRY(hipMemcpy(pri.data(), a->d.prior + base, (size_t)n_nodes * 4, hipMemcpyDeviceToHost));
  }
  const int cells = a->gc.cells, stride = a->gc.m;
  // (the root prints as Black whoever searched: the reference's statefulNode.Player starts as None and ToDot sets it to Black,
  //  graph.go:62-64; the children alternate from there)
  std::vector<int> parent(n_nodes, -1), player(n_nodes, AGZ_BLACK);
  std::string out = "digraph G {\n";
  std::string edges, nodes;
  std::vector<int8_t> board(cells);
  char tmp[160];
  for (int i = 0; i < n_nodes; i++) {
    // this node's board: the moves on the path from the root, each with the colour of its node
    std::fill(board.begin(), board.end(), 0);
    for (int j = i; j >= 0; j = parent[j])
      if (mv[j] >= 0 && mv[j] < cells) board[mv[j]] = (int8_t)player[j];
    nodes += "\t" + std::to_string(i) + " [ fontname=\"Monaco\", shape=none, label=<\n<TABLE BORDER=\"0\" CELLBORDER=\"1\" CELLSPACING=\"0\">\n";
    snprintf(tmp, sizeof tmp, "<TR><TD>Node ID</TD><TD>xx%d</TD></TR>\n<TR><TD>Move</TD><TD>%d</TD></TR>\n", i, (int)mv[i]);
    nodes += tmp;
    snprintf(tmp, sizeof tmp, "<TR><TD>Player</TD><TD>%s</TD></TR>\n<TR><TD>Visits</TD><TD>%u</TD></TR>\n", player[i] == AGZ_BLACK ? "Black" : "White", vis[i]);
    nodes += tmp;
    snprintf(tmp, sizeof tmp, "<TR><TD>Score</TD><TD>%g</TD></TR>\n<TR><TD>Value</TD><TD>-</TD></TR>\n<TR><TD>State</TD><TD>", (double)pri[i]);
    nodes += tmp;
    for (int q = 0; q < cells; q++) {
      if (q % stride == 0) nodes += "\xe2\x8e\xa2 ";                                    // "⎢ "
      nodes += board[q] == AGZ_BLACK ? "X " : (board[q] == AGZ_WHITE ? "O " : "\xc2\xb7 ");  // "·"
      if ((q + 1) % stride == 0 && q != 0) nodes += "\xe2\x8e\xa5<BR />";                // "⎥"
    }
    nodes += "</TD></TR>\n</TABLE>\n> ];\n";
    if (off[i] < 0) continue;
    std::vector<int> kids;
    for (int k = 0; k < kn[i]; k++) if (off[i] + k < n_nodes) kids.push_back(off[i] + k);
    std::sort(kids.begin(), kids.end(), [&](int x, int y) { return mv[x] < mv[y]; });            // sort.Sort(byMove) (graph.go:74)
    for (int c : kids) {
      parent[c] = i;
      player[c] = player[i] == AGZ_BLACK ? AGZ_WHITE : AGZ_BLACK;
      edges += "\t" + std::to_string(i) + "->" + std::to_string(c) + ";\n";
    }
  }
  out += edges + nodes + "}\n";
  *needed = out.size() + 1;
  if (cap) {
    const size_t k = std::min(cap - 1, out.size());
    memcpy(buf, out.data(), k);
    buf[k] = 0;
    m->dot_cache.clear();
  } else {
    m->dot_cache = std::move(out);
  }
  return AGZ_OK;
}

int agz_mcts_nodes(agz_mcts* m, int* n_nodes) {
  AGZ_REQUIRE(m, AGZ_E_INVALID, "mcts is NULL");
  return agz_arena_tree_nodes(m->arena, 0, 0, n_nodes);
}

int agz_mcts_get_stats(agz_mcts* m, agz_arena_stats* out) {
  AGZ_REQUIRE(m, AGZ_E_INVALID, "mcts is NULL");
  return agz_arena_get_stats(m->arena, out);
}

int agz_mcts_reset(agz_mcts* m) {
  AGZ_REQUIRE(m, AGZ_E_INVALID, "mcts is NULL");
  agz_arena* a = m->arena;
  AGZ_REQUIRE(!a->in_move, AGZ_E_STATE, "agz_mcts_reset: a search is in progress");
  AGZ_HIP_TRY(hipSetDevice(a->ctx->device));
  hipStream_t s = a->ctx->stream;
  const int32_t zero = 0;
  for (int t = 0; t < 2; t++) {
    int32_t* f[7] = {a->d.n_nodes, a->d.cur_pool, a->d.has_root, a->d.has_prev, a->d.prev_ply, a->d.stalled, a->d.pc_n};
    for (int i = 0; i < 7; i++) AGZ_HIP_TRY(hipMemcpyAsync(f[i] + t, &zero, 4, hipMemcpyHostToDevice, s));
    AGZ_HIP_TRY(hipMemcpyAsync(a->d.overflow + t, &zero, 4, hipMemcpyHostToDevice, s));
  }
  AGZ_HIP_TRY(hipMemsetAsync(a->d.counters, 0, CNT_N * sizeof(unsigned long long), s));
  AGZ_HIP_TRY(hipStreamSynchronize(s));
  a->prep_expand_seen = 0;
  m->dot_cache.clear(); m->dot_cache.shrink_to_fit();
  m->have_game = false;      // a fresh mcts.New has no game: SetGame comes first
  return AGZ_OK;
}

}  // extern "C"
