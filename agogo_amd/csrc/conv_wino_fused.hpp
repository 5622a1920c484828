// Winograd F(4x4,3x3) dual block with the output transform and the block epilogue INSIDE the GEMM kernel — round 3.
//
// conv_wino_h2.hpp's block moves 2.87 GB through HBM per 512-board launch where the layer's own input and output are 0.38 GB:
// 1.64 GB of it is the fp32 product M = V U written by the GEMM kernel and read back by the output-transform kernel.  Here M
// never exists in memory.  At M A is linear and separable in the position index (xi, nu):
//     Z[l]    += At[l][nu] * M[xi][nu]            after the K loop of position (xi, nu)          4 accumulators per output element
//     Y[k][l] += At[k][xi] * Z[l]                 after the six positions of row xi             16 accumulators per output element
// so a workgroup that owns a (tile block x channel block) and walks all 36 positions keeps 16 + 4 + 2 fp32 registers per
// (tile, GEMM column) and writes y — un-scaled, BN scale/shift, ReLU on both branches, add — and the per-tile maxima the next
// block's input transform reduces.  The register file bounds the workgroup's output tile: 64 tiles x 64 GEMM columns (32 output
// channels, both branches) on 8 waves x 8 accumulators (two waves per SIMD, so one wave's transform arithmetic runs under the
// other's MFMAs).  F(5x5,3x3) would need 25 + 5 + 1 registers per element and does not fit: the fused form runs F(4x4,3x3)
// (900 instead of 784 transform-domain rows per 19x19 board, +15 % MFMA work, no M traffic).
//
// Data path per K step (32 channels): 8 KB of A (64 tiles x 32 k x {hi, lo}) and 8 KB of B (64 columns) arrive by LDS-DMA
// (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write), one 1 KB instruction of each per wave, every instruction
// reading 1 KB of CONTIGUOUS global memory:
//   V  [T/64][36 pos][NK][64 rows][128 B = hi 64 | lo 64]        (written by wino_in_h2_kernel<4> through h.v_row_w / v_chunk_w)
//      a stage's A image is [row][128 B] with the 16-byte segment s of row r stored at slot s ^ (r & 7) — the XOR is applied by
//      the DMA's per-lane source address, the fragment read applies it again: a 16-lane read group touches all 64 banks once;
//   U  [36 pos][NK][Cout_p/32][8 fragments][64 lanes][16 B]      (wino_fused_permute_u2: MFMA operand order, a straight copy).
// Ring of NS stages, ONE raw s_barrier per K step, counted vmcnt (never 0 inside the loop), fragments of step s + 1 read while the
// MFMAs of step s run, the transform arithmetic of a position issued in front of the next position's first MFMA.
//
// Epilogue: the 64 x 64 accumulator tile goes through LDS in two halves ([32 tiles][16 pixels][64 columns] fp32) so that the
// BN parameters are read and y is written in 128-byte runs (8 lanes x float4 per pixel), one wave per tile (its maximum: one
// shuffle reduction, one store).
//
// Roofline: MFMA — 2 * 36 * T * C * Ntot * 3 flop of fp16 MFMA per launch (3 MFMAs per fp32-grade product) against 2516.6 TF.
#pragma once
// (included by net.hip INSIDE namespace agz, after conv_wino_h2.hpp)

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

constexpr int WF_STAGE = 8192;                  // bytes of A (and of B) per K step; the A ring starts at 0, the B ring at WF_BRING
constexpr int WF_BRING = 65536;
constexpr int WF_EP_TILE = 16 * 64 + 4;         // words per tile in the epilogue image (+4: the four tiles one store instruction hits fall on two bank halves)
constexpr int WF_LDS = 32 * WF_EP_TILE * 4;     // 131584 B >= WF_BRING + NS * WF_STAGE

// (PROBE != 0: timing-only variants, results are garbage.  2: no B DMA; 4: no A DMA; 8: no MFMAs / fragment reads; 16: every
//  workgroup reads row block 0 and column block 0 — everything L2-resident; 32: no epilogue)
template <int NK, int NS, int PROBE = 0>
__global__ __launch_bounds__(512, 2) void wino_fused4_h2_kernel(WinoH2Args h) {
  static_assert(NK % NS == 0 && NS >= 3 && NS <= NK && NK % 2 == 0 && NS <= 8, "static stage / register-set indices");
  constexpr int NPOS = 36;
  constexpr int LPS = ((PROBE & 4) ? 0 : 1) + ((PROBE & 2) ? 0 : 1);   // DMA instructions per wave and stage
  __shared__ __attribute__((aligned(16))) unsigned char lds[WF_LDS];
  const WinoArgs& a = h.w;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;               // compute: rows wm*32.., column block wn (16 GEMM columns = 8 channels x 2 branches)

  // workgroup -> (row block of 64 tiles, column block of 32 output channels); an XCD's workgroups are consecutive tiles, so the
  // column blocks that re-read one row block's V run side by side on one L2
  const int n_cb = a.Cout_p >> 5;
  const int n_rb = (a.T + 63) >> 6;
  const int nblk = n_rb * n_cb;
  const int id = blockIdx.x;
  const int q_ = nblk >> 3, rr_ = nblk & 7, xcd = id & 7, slot = id >> 3;
  const int tile = (xcd < rr_ ? xcd * (q_ + 1) : rr_ * (q_ + 1) + (xcd - rr_) * q_) + slot;
  const int rbk = tile / n_cb, cbk = tile - rbk * n_cb;
  const int m0 = rbk * 64, c0 = cbk * 32;

  // LDS-DMA: wave w copies rows 8w .. 8w+7 of the A image (lane = row l/8, slot l%8 <- segment (l%8) ^ (l/8)) and fragment w of B
  const __amdgpu_buffer_rsrc_t rV = h2_rsrc(a.V);
  const __amdgpu_buffer_rsrc_t rU = h2_rsrc(h.U2);
  const unsigned posA = (unsigned)NK * 8192u, posB = (unsigned)NK * (unsigned)n_cb * 8192u, kcB = (unsigned)n_cb * 8192u;
  const unsigned voffA = (unsigned)((PROBE & 16) ? 0 : rbk) * (unsigned)NPOS * posA + (unsigned)(w * 8 + (lane >> 3)) * 128u + (unsigned)(((lane & 7) ^ (lane >> 3)) * 16);
  const unsigned voffB = (unsigned)((PROBE & 16) ? 0 : cbk) * 8192u + (unsigned)w * 1024u + (unsigned)lane * 16u;
  const unsigned dmaA = (unsigned)w * 1024u, dmaB = (unsigned)WF_BRING + (unsigned)w * 1024u;
  // fragment reads: A (row block rb, piece p): row rb*16 + l%16, segment p*4 + l/16, slot = segment ^ (row & 7); B fragment f: f*1024 + l*16
  const unsigned rdA0 = (unsigned)(wm * 32 + (lane & 15)) * 128u + (unsigned)(((lane >> 4) ^ (lane & 7)) * 16);   // block i: + i*2048; lo piece: ^ 64
  const unsigned rdB = (unsigned)WF_BRING + (unsigned)(wn * 2) * 1024u + (unsigned)lane * 16u;

#define WF_ISSUE(ST_, POS_, KC_)                                                                                              \
  {                                                                                                                           \
    if (!(PROBE & 4)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, (lds_ptr_t)(lds + (ST_) * WF_STAGE + dmaA), 16, voffA,      \
                                             (unsigned)(POS_) * posA + (unsigned)(KC_) * 8192u, 0, 0);                        \
    if (!(PROBE & 2)) __builtin_amdgcn_raw_ptr_buffer_load_lds(rU, (lds_ptr_t)(lds + (ST_) * WF_STAGE + dmaB), 16, voffB,      \
                                             (unsigned)(POS_) * posB + (unsigned)(KC_) * kcB, 0, 0);                          \
  }
#define WF_READ(SET_, ST_)                                                                                                    \
  {                                                                                                                           \
    const unsigned char* sb_ = lds + (ST_) * WF_STAGE;                                                                        \
    _Pragma("unroll") for (int p = 0; p < 2; p++) {                                                                           \
      _Pragma("unroll") for (int i = 0; i < 2; i++)                                                                           \
        FA[SET_][i][p] = *reinterpret_cast<const f16x8_t*>(sb_ + ((rdA0 + (unsigned)i * 2048u) ^ ((unsigned)p * 64u)));      \
      FB[SET_][p] = *reinterpret_cast<const f16x8_t*>(sb_ + rdB + (unsigned)p * 1024u);                                       \
    }                                                                                                                         \
  }

  f32x4_t Y[2][4][4], Z[2][4], acc[2];
  f16x8_t FA[2][2][2], FB[2][2];
  const f32x4_t zero4 = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 2; i++) {
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int l = 0; l < 4; l++) Y[i][k][l] = zero4;
#pragma unroll
    for (int l = 0; l < 4; l++) Z[i][l] = zero4;
    acc[i] = zero4;
  }

  // prologue: DMA of steps 0 .. NS-2; stage 0 complete; DMA of step NS-1; fragments of step 0
#pragma unroll
  for (int s = 0; s < NS - 1; s++) WF_ISSUE(s, 0, s)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LPS * (NS - 2)) : "memory");
  __builtin_amdgcn_s_barrier();
  WF_ISSUE(NS - 1, 0, NS - 1)
  if (!(PROBE & 8)) WF_READ(0, 0)

#pragma unroll 1
  for (int xi = 0; xi < 6; xi++) {
#pragma unroll
    for (int nu = 0; nu < 6; nu++) {
      const int pos = xi * 6 + nu;
#pragma unroll
      for (int kc = 0; kc < NK; kc++) {
        // iteration s = pos * NK + kc.  After the counted wait this wave's own DMAs of stage s + 1 have landed and its fragment
        // reads of stage s have returned; after the barrier that holds for every wave: stage s + 1 is complete, stage s is free.
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(LPS * (NS - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        {
          const int kn = kc + NS;                                   // DMA of step s + NS into stage s % NS
          int pn = kn < NK ? pos : pos + 1;
          int kk = kn < NK ? kn : kn - NK;
          if (pn >= NPOS) { pn = NPOS - 1; kk = NK - 1; }           // past the end: a harmless reload keeps the counts uniform
          WF_ISSUE(kc % NS, pn, kk)
        }
        if (!(PROBE & 8)) WF_READ((kc + 1) & 1, (kc + 1) % NS)      // fragments of step s + 1 (past the end: stale bytes, never used)
        __builtin_amdgcn_sched_barrier(0);
        if (kc == 0) {
          // transform of the position that ended with the previous iteration — (xi, nu - 1), or (xi - 1, 5) and the row update of
          // xi - 1 — BEFORE this position's first MFMA overwrites the accumulators (their MFMAs were issued a barrier ago)
          // (At columns: 0: 1,0,0,0 | 1: 1,1,1,1 | 2: 1,-1,1,-1 | 3: 1,2,4,8 | 4: 1,-2,4,-8 | 5: 0,0,0,1)
          if (nu > 0) {
#pragma unroll
            for (int l = 0; l < 4; l++) {
              const float cf = WinoT<4>::AT[l][nu > 0 ? nu - 1 : 0];
              if (cf != 0.f) {
#pragma unroll
                for (int i = 0; i < 2; i++) Z[i][l] += cf * acc[i];
              }
            }
          } else if (xi > 0) {
            const int xp = xi - 1;
#pragma unroll
            for (int i = 0; i < 2; i++) Z[i][3] += acc[i];
            const float ck[4] = {1.f, xp == 1 ? 1.f : xp == 2 ? -1.f : xp == 3 ? 2.f : xp == 4 ? -2.f : 0.f,
                                 (xp == 1 || xp == 2) ? 1.f : (xp == 3 || xp == 4) ? 4.f : 0.f,
                                 xp == 1 ? 1.f : xp == 2 ? -1.f : xp == 3 ? 8.f : xp == 4 ? -8.f : 0.f};   // xp <= 4 here
#pragma unroll
            for (int i = 0; i < 2; i++) {
#pragma unroll
              for (int k = 0; k < 4; k++)
#pragma unroll
                for (int l = 0; l < 4; l++) Y[i][k][l] += ck[k] * Z[i][l];
#pragma unroll
              for (int l = 0; l < 4; l++) Z[i][l] = zero4;
            }
          }
        }
        if (!(PROBE & 8)) {
          const int cs = kc & 1;
#pragma unroll
          for (int pp = 0; pp < 3; pp++) {          // small terms first: lo*hi, hi*lo, hi*hi
            const int pa = pp == 0 ? 1 : 0, pb = pp == 1 ? 1 : 0;
#pragma unroll
            for (int i = 0; i < 2; i++)
              acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FA[cs][i][pa], FB[cs][pb], (kc == 0 && pp == 0) ? zero4 : acc[i], 0, 0, 0);
          }
        }
      }
    }
  }
#undef WF_READ
#undef WF_ISSUE
  // last position (5, 5) and row 5: At[.][5] = (0, 0, 0, 1) in both directions
#pragma unroll
  for (int i = 0; i < 2; i++) {
    Z[i][3] += acc[i];
#pragma unroll
    for (int l = 0; l < 4; l++) Y[i][3][l] += Z[i][l];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (PROBE & 32) {
    if (Y[0][0][0][0] == 12345.678f) h.wave_max[0] = Y[1][3][3][3];   // keep the loop alive
    return;
  }

  // ---- epilogue ----
  // GEMM rows are tile-position-major (row = tt * B + board, h.tile_major): a workgroup's 64 rows are ONE tile position of 64
  // boards (two positions where a row block straddles), so the BN parameters of a lane's items — (pixel, channel quad) of that
  // position — are loaded once and reused for every tile (they were as many bytes as the main loop's operands otherwise).
  // pass i: the 32 tiles {wm*32 + i*16 + 0..15} go to LDS as [tile wm*16 + (lane/16)*4 + r][pixel k*4+l][column wn*16 + lane%16];
  // then wave w owns tiles 4w .. 4w+3 of the pass: lane = (pixel half 8) x (channel quad 8), two iterations per tile.
  float* ef = reinterpret_cast<float*>(lds);
  const int px8 = lane >> 3, cq = lane & 7;                       // pixel within the half, channels 4*cq .. 4*cq+3 of the column block
  const unsigned acol = (unsigned)((cq >> 1) * 16 + (cq & 1) * 4);  // column of branch a (branch b: + 8)
  const float4* ep4 = reinterpret_cast<const float4*>(a.ep) + c0 + 4 * cq;     // {sa, ta, sb, tb} per (pixel, channel)
  float4 E[2][4];
  int tt_cur = -1;
#define WF_EP_LOAD(TT_)                                                                                                       \
  if ((TT_) != tt_cur) {                                                                                                      \
    tt_cur = (TT_);                                                                                                           \
    const int ty_ = tt_cur / a.ntx, tx_ = tt_cur - ty_ * a.ntx;                                                               \
    _Pragma("unroll") for (int hf = 0; hf < 2; hf++) {                                                                        \
      const int px = hf * 8 + px8;                                                                                            \
      const int hh = 4 * ty_ + (px >> 2), ww = 4 * tx_ + (px & 3);                                                            \
      const int hc = hh < a.H ? hh : a.H - 1, wc = ww < a.W ? ww : a.W - 1;                                                   \
      const float4* e = ep4 + (size_t)(hc * a.W + wc) * a.Cout_p;                                                             \
      _Pragma("unroll") for (int u = 0; u < 4; u++) E[hf][u] = e[u];                                                          \
    }                                                                                                                         \
  }
#pragma unroll
  for (int i = 0; i < 2; i++) {
    {
      const int t0 = m0 + ((w * 4) >> 4) * 32 + i * 16 + ((w * 4) & 15);       // this wave's first tile of the pass (uniform)
      const int tc0 = t0 < a.T ? t0 : a.T - 1;
      WF_EP_LOAD(tc0 / a.B)                                        // issued before the LDS round trip
    }
    if (i == 1) __syncthreads();                                   // pass 0's reads are done
    {
      const int tl0 = wm * 16 + (lane >> 4) * 4;
      float* dst = ef + (size_t)tl0 * WF_EP_TILE + wn * 16 + (lane & 15);
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int l = 0; l < 4; l++)
#pragma unroll
          for (int r = 0; r < 4; r++) dst[r * WF_EP_TILE + (k * 4 + l) * 64] = Y[i][k][l][r];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int tl = w * 4 + q;
      const int t = m0 + (tl >> 4) * 32 + i * 16 + (tl & 15);     // GEMM row (uniform)
      const bool tval = t < a.T;
      const int tc = tval ? t : a.T - 1;
      const int tt = tc / a.B, b = tc - tt * a.B;
      WF_EP_LOAD(tt)                                               // (a row block that straddles two tile positions: once more)
      const int ty = tt / a.ntx, tx = tt - ty * a.ntx;
      float s_, un;
      wino_h2_scales(h.amax_in[b], WinoT<4>::VSHIFT, &s_, &un);
      un *= h.w_unscale;
      float* yb = a.y + (size_t)b * a.Hp * a.Wp * a.Cout_p + c0 + 4 * cq;
      float mx = 0.f;
#pragma unroll
      for (int hf = 0; hf < 2; hf++) {
        const int px = hf * 8 + px8;
        const int hh = 4 * ty + (px >> 2), ww = 4 * tx + (px & 3);
        const float* src = ef + (size_t)tl * WF_EP_TILE + px * 64 + acol;
        const float4 ya = *reinterpret_cast<const float4*>(src);
        const float4 yb4 = *reinterpret_cast<const float4*>(src + 8);
        const float va[4] = {ya.x, ya.y, ya.z, ya.w}, vb[4] = {yb4.x, yb4.y, yb4.z, yb4.w};
        float o[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const float4 e = E[hf][u];
          float xa = (va[u] * un) * e.x + e.y;
          float xb = (vb[u] * un) * e.z + e.w;
          xa = xa > 0.f ? xa : 0.f;
          xb = xb > 0.f ? xb : 0.f;
          o[u] = xa + xb;                                         // relu(a) + relu(b) >= 0 already
        }
        if (tval && hh < a.H && ww < a.W) {
          *reinterpret_cast<float4*>(yb + ((size_t)(hh + 1) * a.Wp + (ww + 1)) * a.Cout_p) = make_float4(o[0], o[1], o[2], o[3]);
          mx = fmaxf(fmaxf(mx, fmaxf(o[0], o[1])), fmaxf(o[2], o[3]));
        }
      }
      if (h.wave_max) {
#pragma unroll
        for (int o_ = 32; o_ > 0; o_ >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o_, 64));
        if (lane == 0 && tval) h.wave_max[((size_t)b * a.TPB + tt) * n_cb + cbk] = mx;     // one word per (board, tile, 32 channels)
      }
    }
  }
#undef WF_EP_LOAD
}

// K extents the fused kernel is instantiated for (32-channel steps); the channel count must be a multiple of 32 as well
static inline bool wino_fused_nk_ok(int C) {
  const int nk = C >> 5;
  return (C & 31) == 0 && (nk == 4 || nk == 6 || nk == 8 || nk == 12 || nk == 16);
}
// bytes of V the fused block addresses for `tiles` tiles of C channels
static inline size_t wino_fused_v_bytes(size_t tiles, int C) { return ((tiles + 63) / 64) * 64 * 36 * (size_t)C * 4; }

// Host: conv_wino_h2.hpp's weight image u2[pos][C/32][piece][Ntot][32] -> the fused kernel's [pos][C/32][Cout_p/32][fragment 8][lane 64][8]:
// fragment f = (column block cb = f / 2, piece f % 2), lane l = (GEMM column j = l % 16 of the block, k octet l / 16), column j =
// branch j / 8 of channel cbk*32 + cb*8 + j % 8 — the MFMA B operand of wave column cb, 1 KB per fragment, 8 KB per workgroup and K step.
static void wino_fused_permute_u2(const std::vector<_Float16>& u2, int Cout_p, int C, std::vector<_Float16>& out) {
  const int Ntot = 2 * Cout_p, NC = C / 32, n_cb = Cout_p / 32;
  out.assign(u2.size(), (_Float16)0.f);
  for (int pos = 0; pos < 36; pos++)
    for (int kc = 0; kc < NC; kc++)
      for (int cbk = 0; cbk < n_cb; cbk++)
        for (int f = 0; f < 8; f++)
          for (int l = 0; l < 64; l++) {
            const int cb = f >> 1, p = f & 1, j = l & 15, q = l >> 4;
            const int n = (j >> 3) * Cout_p + cbk * 32 + cb * 8 + (j & 7);
            const _Float16* src = &u2[((((size_t)pos * NC + kc) * 2 + p) * Ntot + n) * 32 + q * 8];
            _Float16* dst = &out[(((((size_t)pos * NC + kc) * n_cb + cbk) * 8 + f) * 64 + l) * 8];
            for (int e = 0; e < 8; e++) dst[e] = src[e];
          }
}

// input transform F(4x4,3x3) + the fused kernel for one dual block (h.w.V: wino_fused_v_bytes(); h.U2: wino_fused_permute_u2's image)
static void wino_fused_launch(agz_ctx* ctx, WinoH2Args& h, hipStream_t st = nullptr) {
  if (!st) st = ctx->stream;
  WinoArgs& a = h.w;
  const int nk = a.C >> 5;
  h.tm = 4; h.npos = 36;
  a.nty = ceil_div(a.H, 4); a.ntx = ceil_div(a.W, 4); a.TPB = a.nty * a.ntx; a.T = a.B * a.TPB;
  // V rows: (t / 64) * (36 * NK * 64) + pos * (NK * 64) + t % 64, 32 words each, chunks 64 rows apart
  h.rsh = 6; h.rmask = 63; h.rA = 36u * (unsigned)nk * 64u; h.rB = (unsigned)nk * 64u;
  h.v_row_w = 32u; h.v_chunk_w = 64u * 32u; h.tile_major = 1;
  a.n_mtiles = ceil_div(a.T, 128); a.n_ntiles = ceil_div(a.Ntot, 128);
  h.in_swap = a.C % 128 == 0 ? 1 : 0;
  const bool fuse = a.C % 128 == 0 && h.wave_max != nullptr;
  h.wm_per_board = a.TPB * (a.Cout_p >> 5);                           // wave_max: one word per tile and 32 channels
  h.fuse_prev = (h.fuse_prev && fuse) ? 1 : 0;
  h.amax_self = const_cast<unsigned*>(h.amax_in);
  h.raw = 0;
  {
    ProfScopeOn ps(ctx, AGZ_PROF_WINO_IN, st == ctx->stream);
    const size_t n_in = (size_t)a.T * (a.C / 2);
    hipLaunchKernelGGL(wino_in_h2_kernel<4>, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, st, h);
  }
  {
    ProfScopeOn ps(ctx, AGZ_PROF_WINO_GEMM, st == ctx->stream);
    const dim3 grid((unsigned)(ceil_div(a.T, 64) * (a.Cout_p >> 5)));
    // timing-only probes of the K = 256 kernel (AGZ_WINO_H2_FUSED_PROBE, see the kernel header): measurement knob
    static const int probe = [] { const char* e = getenv("AGZ_WINO_H2_FUSED_PROBE"); return e ? atoi(e) : 0; }();
#define AGZ_WF(NK_, NS_, P_) hipLaunchKernelGGL((wino_fused4_h2_kernel<NK_, NS_, P_>), grid, dim3(512), 0, st, h)
    if (nk == 8 && probe) {
      switch (probe) {
        case 2: AGZ_WF(8, 8, 2); break;
        case 4: AGZ_WF(8, 8, 4); break;
        case 6: AGZ_WF(8, 8, 6); break;
        case 8: AGZ_WF(8, 8, 8); break;
        case 16: AGZ_WF(8, 8, 16); break;
        case 24: AGZ_WF(8, 8, 24); break;
        case 32: AGZ_WF(8, 8, 32); break;
        case 38: AGZ_WF(8, 8, 38); break;
        case 40: AGZ_WF(8, 8, 40); break;
        case 99: AGZ_WF(8, 4, 0); break;      // 4-stage ring
        default: AGZ_WF(8, 8, 0); break;
      }
    }
    else if (nk == 8) AGZ_WF(8, 8, 0);
    else if (nk == 4) AGZ_WF(4, 4, 0);
    else if (nk == 16) AGZ_WF(16, 8, 0);
    else if (nk == 12) AGZ_WF(12, 4, 0);
    else AGZ_WF(6, 3, 0);
#undef AGZ_WF
  }
  if (h.wave_max && h.amax_out && !fuse)
    hipLaunchKernelGGL(wino_board_max_kernel, dim3(a.B), dim3(64), 0, st, h.wave_max, h.amax_out, h.wm_per_board);
}
