// Winograd F(5x5,3x3) / F(4x4,3x3) dual block with fp16x2 products in the transform domain — AGZ_COMPUTE_WINO_H2.
//
// conv_wino.hpp's block spends 0.68 of its 1.07 ms in 36 GEMMs whose bf16x3 products cost six MFMAs each and whose staging
// splits every fp32 activation into three pieces on the VALU.  Here the transform-domain operands are written ALREADY SPLIT:
//     V = hi + lo,  hi = RN_fp16(V*s),  lo = RN_fp16(V*s - hi)          (s: a power of two per BOARD, below)
// by the input-transform kernel — 4 bytes per element, exactly what the fp32 V took — so the GEMM's staging is a pure copy
// (no VALU), and a product is three fp16 MFMAs (hi*hi, hi*lo, lo*hi; the dropped lo*lo is <= 2^-22 relative): half the
// matrix instructions of the bf16x3 form, none of its split arithmetic.
//
// Range management.  fp16 has 5 exponent bits, so each operand is scaled by a power of two (exact):
//   weights  U = G g Gt: per layer, max|U| * su in [2^13, 2^14)  — fixed at commit (inference) or computed on the device
//   from the current filter (training, wino_u_*_kernel);
//   V = Bt d B of board b: |V| <= (max row sum of |Bt|)^2 * max|d| <= 2^VSHIFT max|d| (100 for F(4x4), 56.25 for F(5x5)), so with
//   amax_b = max |x| over the board's layer input, s_b = 2^(141 - VSHIFT - E(amax_b)) puts every |V*s_b| below 2^15 (fp16 max
//   65504): overflow is impossible, and an element hi+lo carries an ABSOLUTE error <= 2^-25 in scaled units, i.e. <= 2^-38 of the
//   largest representable |V| — far below fp32's own 2^-24 relative rounding of the accumulated sums.  amax_b is a per-board
//   quantity — every wave of the producing output transform stores the maximum of what it wrote, the consuming input transform
//   reduces its board's words — so a board's result does not depend on what else is in the batch, bit for bit.
//   M = V U comes out scaled by s_b*su and is un-scaled (exactly) after the output transform.
//
// Kernels per block:   wino_in_h2_kernel (x -> V2, HBM-bound) ; wino_gemm_h2d_kernel (HBM/MFMA; plain fallbacks for other K
// extents) ; wino_out_seq_h2_kernel / wino_out_h2_kernel (M -> y + per-wave maxima of y for the next block, HBM-bound) ;
// training: wino_out_raw_h2_kernel.  Layouts (blocked, default): V2[T/128][pos][128 rows][C/32][piece 2][32] fp16 (a row's K
// range is contiguous: 128 B per 32-channel chunk, hi then lo);  U2[pos][C/32][piece 2][Ntot][32] fp16;
// M[T/128][pos][128 rows][Ntot] fp32.  Measurements and the history of each kernel: DESIGN.md section 4e.
#pragma once
// (included by net.hip INSIDE namespace agz, after conv_wino.hpp and conv_h2.hpp)

// Transform matrices (Cook-Toom; the last interpolation point is infinity):
//   TM = 4: F(4x4,3x3), points 0, +-1, +-2      (Lavin & Gray 2016; conv_wino.hpp)        6x6 = 36 positions per 16 outputs
//   TM = 5: F(5x5,3x3), points 0, +-1, +-1/2, 2                                          7x7 = 49 positions per 25 outputs
// A 19x19 board is 5x5 tiles of 4 (900 GEMM rows) or 4x4 tiles of 5 (784 rows: V, M and the GEMMs shrink by 13 %); a 9x9 board
// 3x3 tiles of 4 (324 rows, 78 % overhang) or 2x2 tiles of 5 (196 rows).  One layer's rounding error on post-ReLU data (fp32
// transforms, C = 256): 1.4e-6 of the output rms for TM 4, 2.3e-6 for TM 5 (direct fp32 accumulation: 2.1e-7) — both two orders
// inside the network tolerance.  |Bt| row sums: 10 (TM 4), 7.5 (TM 5) -> |V| <= 100 / 56.25 max|d|: the range bound below.
template <int TM> struct WinoT;
template <> struct WinoT<4> {
  static constexpr int AL = 6, VSHIFT = 7;    // |V| <= 2^VSHIFT * max|d|
  static constexpr float BT[6][6] = {{4, 0, -5, 0, 1, 0}, {0, -4, -4, 1, 1, 0}, {0, 4, -4, -1, 1, 0}, {0, -2, -1, 2, 1, 0}, {0, 2, -1, -2, 1, 0}, {0, 4, 0, -5, 0, 1}};
  static constexpr float AT[4][6] = {{1, 1, 1, 1, 1, 0}, {0, 1, -1, 2, -2, 0}, {0, 1, 1, 4, 4, 0}, {0, 1, -1, 8, -8, 1}};
  static constexpr double G[6][3] = {{1.0 / 4, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6}, {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
};
template <> struct WinoT<5> {
  static constexpr int AL = 7, VSHIFT = 6;
  static constexpr float BT[7][7] = {{-0.5f, 0.25f, 2.5f, -1.25f, -2, 1, 0}, {0, 0.5f, 0.25f, -2.25f, -1, 1, 0}, {0, -0.5f, 0.75f, 1.75f, -3, 1, 0},
                                     {0, 1, 1.5f, -2, -1.5f, 1, 0},          {0, -1, 2.5f, 0, -2.5f, 1, 0},       {0, 0.25f, 0, -1.25f, 0, 1, 0},
                                     {0, -0.5f, 0.25f, 2.5f, -1.25f, -2, 1}};
  static constexpr float AT[5][7] = {{1, 1, 1, 1, 1, 1, 0}, {0, 1, -1, 0.5f, -0.5f, 2, 0}, {0, 1, 1, 0.25f, 0.25f, 4, 0}, {0, 1, -1, 0.125f, -0.125f, 8, 0},
                                     {0, 1, 1, 0.0625f, 0.0625f, 16, 1}};
  static constexpr double G[7][3] = {{-2, 0, 0}, {-2.0 / 3, -2.0 / 3, -2.0 / 3}, {-2.0 / 9, 2.0 / 9, -2.0 / 9}, {16.0 / 9, 8.0 / 9, 4.0 / 9}, {16.0 / 15, -8.0 / 15, 4.0 / 15},
                                     {2.0 / 45, 4.0 / 45, 8.0 / 45}, {0, 0, 1}};
};
// the 1-D transforms written out (shared sub-sums; a generic matrix-vector loop over the constexpr tables above compiled to 35 %
// slower transform kernels: measured 0.19 vs 0.14 ms for the TM = 4 input transform)
template <int TM> __device__ __forceinline__ void wino_btv(const float* d, float* o);
template <int TM> __device__ __forceinline__ void wino_atv(const float* m, float* o);
template <> __device__ __forceinline__ void wino_btv<4>(const float* d, float* o) { wino_bt6(d, o); }
template <> __device__ __forceinline__ void wino_atv<4>(const float* m, float* o) { wino_at4(m, o); }
template <> __device__ __forceinline__ void wino_btv<5>(const float* d, float* o) {
  o[0] = -0.5f * d[0] + 0.25f * d[1] + 2.5f * d[2] - 1.25f * d[3] - 2.f * d[4] + d[5];
  o[1] = 0.5f * d[1] + 0.25f * d[2] - 2.25f * d[3] - d[4] + d[5];
  o[2] = -0.5f * d[1] + 0.75f * d[2] + 1.75f * d[3] - 3.f * d[4] + d[5];
  o[3] = d[1] + 1.5f * d[2] - 2.f * d[3] - 1.5f * d[4] + d[5];
  o[4] = -d[1] + 2.5f * (d[2] - d[4]) + d[5];
  o[5] = 0.25f * d[1] - 1.25f * d[3] + d[5];
  o[6] = -0.5f * d[1] + 0.25f * d[2] + 2.5f * d[3] - 1.25f * d[4] - 2.f * d[5] + d[6];
}
template <> __device__ __forceinline__ void wino_atv<5>(const float* m, float* o) {
  const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  o[0] = m[0] + s12 + s34 + m[5];
  o[1] = d12 + 0.5f * d34 + 2.f * m[5];
  o[2] = s12 + 0.25f * s34 + 4.f * m[5];
  o[3] = d12 + 0.125f * d34 + 8.f * m[5];
  o[4] = s12 + 0.0625f * s34 + 16.f * m[5] + m[6];
}

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
struct WinoH2Args;
__device__ __forceinline__ size_t h2_row(const WinoH2Args& h, int pos, int t);
struct WinoH2Args {
  WinoArgs w;                // geometry (nty/ntx/TPB/T in tiles of tm), x, V (reinterpreted as the fp16 piece image), Mb, ep, y
  int tm, npos;              // tile size 4 or 5; (tm + 2)^2 positions
  // Row of (position, tile t) in V and M = (t >> rsh) * rA + pos * rB + (t & rmask), set by wino_h2_launch.  Blocked layout
  // (default): [T/128][npos][128 rows] — the 128-row tile a GEMM block produces is contiguous and a tile's positions lie
  // within npos * 128 rows, so the transform kernels' (tm+2)^2 streams stay inside a few pages.  Plain layout: [npos][T + pad].
  int rsh, rmask;
  unsigned rA, rB;
  int raw;                   // 1 (training convolutions): the output stage is wino_out_raw_h2_kernel — At M A un-scaled, no epilogue,
                             // all Ntot columns as they are, y [B][Hp][Wp][Ntot] — and w_unscale is read from w_unscale_dev
  const float* w_unscale_dev;
  int fuse_prev;             // 1: this block's input range is reduced by the input transform itself from wave_max (what the
                             // previous block's output transform left there) and written to amax_self; 0: amax_in is ready
  int wm_per_board;          // words of wave_max per board (set by wino_h2_launch)
  unsigned* amax_self;       // = amax_in, writable
  int in_swap;               // input transform stores 256-byte runs through lane swaps (C % 128 == 0; tuning knob AGZ_WINO_H2_IN_SWAP)
  const _Float16* U2;        // [36][C/32][2][Ntot][32]
  const unsigned* amax_in;   // [B] max |x| of every board of this block's input (float bits)
  unsigned* amax_out;        // [B] max of this block's output (wino_board_max_kernel over wave_max)
  float* wave_max;           // [T][Cout_p/64] maximum of the 64 channels x 16 pixels one wave of wino_out_h2_kernel produced
  float w_unscale;           // 1 / su (one power of two for the whole layer: the device-built training image)
  // Equilibrated inference image (wino_build_u2): input channel ci of U is divided by the power of two t_in[ci] and V's channel
  // ci multiplied by it (exact, the products do not change); every GEMM column n has its own power-of-two scale, col_unscale[n]
  // = 1 / su_n.  A layer whose channels / filters differ by many octaves then keeps fp32-grade pieces everywhere.
  // nullptr: no channel scaling / the scalar w_unscale.  t_next = the NEXT block's t_in: the per-wave maxima this block's output
  // transform records are maxima of y * t_next (what the next input transform's range bound needs).
  const float* t_in;
  const float* col_unscale;
  const float* t_next;
  // ---- chained form (conv_wino_h2c.hpp): V2c[T/128][pos][C/32][128 rows][hi 64 B | lo 64 B], Mc[T/128][pos][C/32 slices][128 rows][64 cols]
  // (column 2 j + branch of slice s = channel 32 s + j), U2c[pos][C/32][Ntot/256][256 cols][hi 64 B | lo 64 B].
  int cform;                 // 1: wino_in_h2_kernel stores V2c
  const _Float16* U2c;
  const float* amax_true;    // [B] exact max |x| of this block's input (float bits) — block 0; nullptr: reduce wm_prev
  const float* wm_prev;      // [B][TPB * C/32] per-(tile, slice) maxima of this block's input, left by the kernel that produced it
  float* wm_out;             // the same for this block's output (the other half of the ping-pong pair)
  unsigned* amax_next;       // [B] bits of the proven bound on max |y| of this block's output: the range word of V2(l+1)
  float g1, g0;              // that bound = g1 * max|x| + g0 (agz_net::build_wino_h2_weights)
  int gemm_variant;          // 0: default, 1: wino_gemm_h2g_kernel, 2: wino_gemm_h2p_kernel (agz_net_set_wino_h2_gemm, agz_debug.h)
  int temporal_stores;       // A/B (agz_net_set_wino_h2_gemm + 64): 1 = M and V2c stored with the default cache policy (round 4); 0: non-temporal
  int row_pad;               // three-kernel form: extra rows after every position's 128 tile rows of V and M (wino_h2_launch: rB = 128 + row_pad)
};
__device__ __forceinline__ size_t h2_row(const WinoH2Args& h, int pos, int t) {
  return (size_t)(t >> h.rsh) * h.rA + (size_t)pos * h.rB + (size_t)(t & h.rmask);
}

// s = 2^(141 - shift - E): |V| <= 2^shift * amax < 2^(E - 126 + shift)  =>  |V * s| < 2^15.   inv = 1 / s.   (shift 7: 134 - E)
__device__ __forceinline__ void wino_h2_scales(unsigned amax_bits, int shift, float* s, float* inv) {
  int e = (int)((amax_bits >> 23) & 0xffu);
  if (amax_bits == 0u) { *s = 1.f; *inv = 1.f; return; }
  e = e < 30 ? 30 : (e > 230 ? 230 : e);
  *s = __uint_as_float((unsigned)(268 - shift - e) << 23);
  *inv = __uint_as_float((unsigned)(e - 14 + shift) << 23);
}

__device__ __forceinline__ unsigned wino_h2_pack(float a, float b, unsigned* lo) {
  const _Float16 ha = (_Float16)a, hb = (_Float16)b;
  const _Float16 la = (_Float16)(a - (float)ha), lb = (_Float16)(b - (float)hb);
  *lo = (unsigned)__builtin_bit_cast(unsigned short, la) | ((unsigned)__builtin_bit_cast(unsigned short, lb) << 16);
  return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
}

// One thread per (tile, channel pair) like wino_in_kernel; the result is scaled by the board's power of two, split, and
// stored as one 4-byte hi word and one 4-byte lo word (16 lanes fill the 64-byte hi / lo halves of a 32-channel chunk).
template <int TM>
__global__ __launch_bounds__(256) void wino_in_h2_kernel(WinoH2Args h) {
  using WT = WinoT<TM>;
  constexpr int AL = WT::AL;
  const WinoArgs& a = h.w;
  const int C2 = a.C >> 1;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (size_t)a.T * C2) return;
  const int c2 = (int)(g % C2);
  const int t = (int)(g / C2);
  const int b = t / a.TPB, tt = t - b * a.TPB;
  const int ty = tt / a.ntx, tx = tt - ty * a.ntx;
  unsigned amax_bits;
  if (h.fuse_prev) {   // (uniform; C % 128 == 0: the wave's 64 lanes are one tile of one board)
    // the board's range = max over the per-wave maxima the previous block's output transform stored: 64..128 words, L2-resident;
    // every wave reduces them itself (this replaces a one-wave-per-board kernel between every two blocks)
    const float* wp = h.wave_max + (size_t)b * h.wm_per_board;
    float mx = 0.f;
    for (int i = threadIdx.x & 63; i < h.wm_per_board; i += 64) mx = fmaxf(mx, wp[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    amax_bits = __float_as_uint(mx);
    if (tt == 0 && c2 == 0) h.amax_self[b] = amax_bits;   // for this block's output transform
  } else {
    amax_bits = h.amax_in[b];
  }
  float sb, inv_;
  wino_h2_scales(amax_bits, WT::VSHIFT, &sb, &inv_);
  // (amax_bits bounds |x * t_in| over the board, so |V * sb * t| < 2^15 for every channel)
  float sbx = sb, sby = sb;
  if (h.t_in) { const float2 t2 = *reinterpret_cast<const float2*>(h.t_in + 2 * c2); sbx *= t2.x; sby *= t2.y; }
  const float* xb = a.x + (size_t)b * a.Hp * a.Wp * a.C + 2 * c2;
  // all loads are issued unconditionally (clamped address, zeroed afterwards): a branch per load keeps only one column in
  // flight (measured 4.2 TB/s of algorithmic bytes with the branches)
  float2 d[AL][AL];
#pragma unroll
  for (int j = 0; j < AL; j++) {
    const int px = TM * tx + j, pxc = px < a.Wp ? px : a.Wp - 1;
#pragma unroll
    for (int i = 0; i < AL; i++) {
      const int py = TM * ty + i, pyc = py < a.Hp ? py : a.Hp - 1;
      d[i][j] = *reinterpret_cast<const float2*>(xb + ((size_t)pyc * a.Wp + pxc) * a.C);
    }
  }
  float tmx[AL][AL], tmy[AL][AL];
#pragma unroll
  for (int j = 0; j < AL; j++) {
    const bool okx = TM * tx + j < a.Wp;
    float dx[AL], dy[AL], ox[AL], oy[AL];
#pragma unroll
    for (int i = 0; i < AL; i++) {
      const bool ok = okx && TM * ty + i < a.Hp;
      dx[i] = ok ? d[i][j].x : 0.f; dy[i] = ok ? d[i][j].y : 0.f;
    }
    wino_btv<TM>(dx, ox);
    wino_btv<TM>(dy, oy);
#pragma unroll
    for (int i = 0; i < AL; i++) { tmx[i][j] = ox[i]; tmy[i][j] = oy[i]; }
  }
  unsigned* V2 = reinterpret_cast<unsigned*>(a.V);          // 4-byte words: [pos][T][C/32][2][16]
  const int c = 2 * c2;
  const size_t word_in_row = (size_t)(c >> 5) * 32 + ((c & 31) >> 1);   // hi word; the lo word sits 16 words further
  // Store forms (h.in_swap, uniform).  0: every lane stores its own hi and lo word — per instruction four 64-byte runs (16 lanes
  // fill the hi half of a 32-channel chunk).  1 (C % 128 == 0, the block size keeps a wave inside one tile): v_permlane16_swap
  // gathers a chunk's hi and lo halves into 32 neighbouring lanes and v_permlane32_swap puts two neighbouring chunks into one
  // register: per instruction ONE 256-byte run (lanes 0..63 = 64 consecutive words of the row).
  const int lane = threadIdx.x & 63;
  const size_t swap_word = (size_t)((c2 >> 6) * 4) * 32 + lane;           // chunks 4q, 4q+1 of the wave's 4 chunks; the others 64 words on
#pragma unroll
  for (int i = 0; i < AL; i++) {
    float ox[AL], oy[AL];
    wino_btv<TM>(tmx[i], ox);
    wino_btv<TM>(tmy[i], oy);
#pragma unroll
    for (int j = 0; j < AL; j++) {
      unsigned lo;
      const unsigned hi = wino_h2_pack(ox[j] * sbx, oy[j] * sby, &lo);
      if (h.cform) {   // (uniform; C % 128 == 0) chained layout: the wave's four K steps are four 16 KB chunks — two 128-byte row runs per store
        const auto s16 = __builtin_amdgcn_permlane16_swap(hi, lo, false, false);   // [hi k0, lo k0, hi k2, lo k2], [hi k1, lo k1, hi k3, lo k3]
        const unsigned e0 = s16[0], e1 = s16[1];
        const int NS = a.C >> 5;
        const size_t chunk0 = ((size_t)(t >> 7) * h.npos + (size_t)(i * AL + j)) * NS + (size_t)((c2 >> 6) * 4 + 2 * (lane >> 5));   // K step of e0's half
        unsigned* p0 = V2 + chunk0 * 4096 + (size_t)(t & 127) * 32 + (lane & 31);
        p0[0] = e0;
        p0[4096] = e1;                                                              // the next K step's chunk
        continue;
      }
      unsigned* rowp = V2 + h2_row(h, i * AL + j, t) * a.C;
      if (h.in_swap) {
        const auto s16 = __builtin_amdgcn_permlane16_swap(hi, lo, false, false);   // [hi r0, lo r0, hi r2, lo r2], [hi r1, lo r1, hi r3, lo r3]
        const unsigned e0 = s16[0], e1 = s16[1];
        const auto s32 = __builtin_amdgcn_permlane32_swap(e0, e1, false, false);   // [chunk 0, chunk 1], [chunk 2, chunk 3]
        const unsigned w0 = s32[0], w1 = s32[1];
        rowp[swap_word] = w0;
        rowp[swap_word + 64] = w1;
      } else {
        rowp[word_in_row] = hi;
        rowp[word_in_row + 16] = lo;
      }
    }
  }
}

// The 36 GEMMs with fp16x2 products, 128 x 128 tile: conv3x3_h2_kernel's tile / LDS image / single-stage pipeline with a
// plain K loop over 32-channel chunks; the A operand arrives already split (pure copy into LDS).
__global__ __launch_bounds__(256, 3) void wino_gemm_h2_kernel(WinoH2Args h) {
  const WinoArgs& a = h.w;
  constexpr int PIECE = 128 * 64;
  constexpr int STAGE = 4 * PIECE;           // A hi, A lo, B hi, B lo
  __shared__ __attribute__((aligned(16))) unsigned char lds[STAGE];

  const int per_pos = a.n_mtiles * a.n_ntiles;
  const int nblk = h.npos * per_pos;
  const int id = blockIdx.x;
  int q = nblk >> 3, rr = nblk & 7, xcd = id & 7, slot = id >> 3;
  int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + slot;
  const int pos = tile / per_pos;
  const int rem = tile - pos * per_pos;
  const int m_tile = rem / a.n_ntiles, n_tile = rem - m_tile * a.n_ntiles;
  const int m0 = m_tile * 128, n0 = n_tile * 128;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int srow = tid >> 1, shalf = tid & 1;
  int mrow = m0 + srow;
  if (mrow >= a.T) mrow = a.T - 1;
  int nrow = n0 + srow;
  if (nrow >= a.Ntot) nrow = a.Ntot - 1;
  const int NK = a.C >> 5;
  const unsigned piece_bytes = (unsigned)a.Ntot * 64u;
  unsigned xo_ = (unsigned)((h2_row(h, pos, mrow) * a.C) * 4) + (unsigned)shalf * 32u;
  unsigned wo_ = (unsigned)pos * (unsigned)NK * 2u * piece_bytes + (unsigned)nrow * 64u + (unsigned)shalf * 32u;
  const unsigned s_off0 = h2_lds_off(srow, 2 * shalf), s_off1 = h2_lds_off(srow, 2 * shalf + 1);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  int ra[2], rb[2];
#pragma unroll
  for (int i = 0; i < 2; i++) ra[i] = (wm * 2 + i) * 32 + (lane & 31);
#pragma unroll
  for (int j = 0; j < 2; j++) rb[j] = (wn * 2 + j) * 32 + (lane & 31);
  const int kh = lane >> 5;

  u32x4_t xa0, xa1, xa2, xa3, xb0, xb1, xb2, xb3;
  const char* xbase = reinterpret_cast<const char*>(a.V);
  const char* wbase = reinterpret_cast<const char*>(h.U2);
  int f_n = 0;
#define WH2_GLOAD()                                                                           \
  xa0 = *reinterpret_cast<const u32x4_t*>(xbase + xo_);                                       \
  xa1 = *reinterpret_cast<const u32x4_t*>(xbase + xo_ + 16u);                                 \
  xa2 = *reinterpret_cast<const u32x4_t*>(xbase + xo_ + 64u);                                 \
  xa3 = *reinterpret_cast<const u32x4_t*>(xbase + xo_ + 80u);                                 \
  xb0 = *reinterpret_cast<const u32x4_t*>(wbase + wo_);                                       \
  xb1 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + 16u);                                 \
  xb2 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + piece_bytes);                         \
  xb3 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + piece_bytes + 16u);                   \
  if (f_n + 1 < NK) { f_n++; xo_ += 128u; wo_ += 2u * piece_bytes; }
#define WH2_STORE()                                                                           \
  *reinterpret_cast<u32x4_t*>(lds + 0 * PIECE + s_off0) = xa0;                                \
  *reinterpret_cast<u32x4_t*>(lds + 0 * PIECE + s_off1) = xa1;                                \
  *reinterpret_cast<u32x4_t*>(lds + 1 * PIECE + s_off0) = xa2;                                \
  *reinterpret_cast<u32x4_t*>(lds + 1 * PIECE + s_off1) = xa3;                                \
  *reinterpret_cast<u32x4_t*>(lds + 2 * PIECE + s_off0) = xb0;                                \
  *reinterpret_cast<u32x4_t*>(lds + 2 * PIECE + s_off1) = xb1;                                \
  *reinterpret_cast<u32x4_t*>(lds + 3 * PIECE + s_off0) = xb2;                                \
  *reinterpret_cast<u32x4_t*>(lds + 3 * PIECE + s_off1) = xb3;
#define WH2_QUAD(PA_, PB_)                                                                    \
  _Pragma("unroll") for (int i = 0; i < 2; i++)                                               \
    _Pragma("unroll") for (int j = 0; j < 2; j++)                                             \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[i][PA_], B_[j][PB_], acc[i][j], 0, 0, 0);

  WH2_GLOAD()
  for (int it = 0; it < NK; it++) {
    WH2_STORE()
    __syncthreads();
    WH2_GLOAD()
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      f16x8_t A_[2][2], B_[2][2];
#pragma unroll
      for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int i = 0; i < 2; i++) A_[i][p] = *reinterpret_cast<const f16x8_t*>(lds + p * PIECE + h2_lds_off(ra[i], 2 * ks + kh));
#pragma unroll
        for (int j = 0; j < 2; j++) B_[j][p] = *reinterpret_cast<const f16x8_t*>(lds + (2 + p) * PIECE + h2_lds_off(rb[j], 2 * ks + kh));
      }
      WH2_QUAD(1, 0) WH2_QUAD(0, 1) WH2_QUAD(0, 0)   // small terms first
    }
    __syncthreads();
  }
#undef WH2_QUAD
#undef WH2_STORE
#undef WH2_GLOAD

#pragma unroll
  for (int i = 0; i < 2; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int m = m0 + row;
      if (m < a.T) {
        float* dst = a.Mb + h2_row(h, pos, m) * a.Ntot;
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int c = n0 + (wn * 2 + j) * 32 + (lane & 31);
          if (c < a.Ntot) dst[c] = acc[i][j][r];
        }
      }
    }
  }
}

// Wide variant: 128 x 256 tile, 64 x 128 per wave (conv3x3_h2w_kernel's shape): half the A re-reads and LDS traffic per MFMA.
__global__ __launch_bounds__(256, 2) void wino_gemm_h2w_kernel(WinoH2Args h) {
  const WinoArgs& a = h.w;
  constexpr int PA = 128 * 64, PB = 256 * 64;
  constexpr int STAGE = 2 * PA + 2 * PB;          // 48 KB
  __shared__ __attribute__((aligned(16))) unsigned char lds[STAGE];

  const int n_nt = (a.Ntot + 255) / 256;
  const int per_pos = a.n_mtiles * n_nt;
  const int nblk = h.npos * per_pos;
  const int id = blockIdx.x;
  int q = nblk >> 3, rr = nblk & 7, xcd = id & 7, slot = id >> 3;
  int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + slot;
  const int pos = tile / per_pos;
  const int rem = tile - pos * per_pos;
  const int m_tile = rem / n_nt, n_tile = rem - m_tile * n_nt;
  const int m0 = m_tile * 128, n0 = n_tile * 256;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int srow = tid >> 1, shalf = tid & 1;
  int mrow = m0 + srow;
  if (mrow >= a.T) mrow = a.T - 1;
  int nr0 = n0 + srow, nr1 = n0 + 128 + srow;
  if (nr0 >= a.Ntot) nr0 = a.Ntot - 1;
  if (nr1 >= a.Ntot) nr1 = a.Ntot - 1;
  const int NK = a.C >> 5;
  const unsigned piece_bytes = (unsigned)a.Ntot * 64u;
  unsigned xo_ = (unsigned)((h2_row(h, pos, mrow) * a.C) * 4) + (unsigned)shalf * 32u;
  unsigned wo_ = (unsigned)pos * (unsigned)NK * 2u * piece_bytes;
  const unsigned b_g0 = (unsigned)nr0 * 64u + (unsigned)shalf * 32u, b_g1 = (unsigned)nr1 * 64u + (unsigned)shalf * 32u;
  const unsigned sa0 = h2_lds_off(srow, 2 * shalf), sa1 = h2_lds_off(srow, 2 * shalf + 1);
  const unsigned sb0 = h2_lds_off(128 + srow, 2 * shalf), sb1 = h2_lds_off(128 + srow, 2 * shalf + 1);

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  int ra[2], rb[4];
#pragma unroll
  for (int i = 0; i < 2; i++) ra[i] = (wm * 2 + i) * 32 + (lane & 31);
#pragma unroll
  for (int j = 0; j < 4; j++) rb[j] = wn * 128 + j * 32 + (lane & 31);
  const int kh = lane >> 5;

  u32x4_t xa0, xa1, xa2, xa3, xb0, xb1, xb2, xb3, xb4, xb5, xb6, xb7;
  const char* xbase = reinterpret_cast<const char*>(a.V);
  const char* wbase = reinterpret_cast<const char*>(h.U2);
  int f_n = 0;
#define WH2W_GLOAD()                                                                          \
  xa0 = *reinterpret_cast<const u32x4_t*>(xbase + xo_);                                       \
  xa1 = *reinterpret_cast<const u32x4_t*>(xbase + xo_ + 16u);                                 \
  xa2 = *reinterpret_cast<const u32x4_t*>(xbase + xo_ + 64u);                                 \
  xa3 = *reinterpret_cast<const u32x4_t*>(xbase + xo_ + 80u);                                 \
  xb0 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + b_g0);                                \
  xb1 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + b_g0 + 16u);                          \
  xb2 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + piece_bytes + b_g0);                  \
  xb3 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + piece_bytes + b_g0 + 16u);            \
  xb4 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + b_g1);                                \
  xb5 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + b_g1 + 16u);                          \
  xb6 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + piece_bytes + b_g1);                  \
  xb7 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + piece_bytes + b_g1 + 16u);            \
  if (f_n + 1 < NK) { f_n++; xo_ += 128u; wo_ += 2u * piece_bytes; }
#define WH2W_STORE()                                                                          \
  {                                                                                           \
    *reinterpret_cast<u32x4_t*>(lds + 0 * PA + sa0) = xa0;                                    \
    *reinterpret_cast<u32x4_t*>(lds + 0 * PA + sa1) = xa1;                                    \
    *reinterpret_cast<u32x4_t*>(lds + 1 * PA + sa0) = xa2;                                    \
    *reinterpret_cast<u32x4_t*>(lds + 1 * PA + sa1) = xa3;                                    \
    unsigned char* sb_ = lds + 2 * PA;                                                        \
    *reinterpret_cast<u32x4_t*>(sb_ + 0 * PB + sa0) = xb0;                                    \
    *reinterpret_cast<u32x4_t*>(sb_ + 0 * PB + sa1) = xb1;                                    \
    *reinterpret_cast<u32x4_t*>(sb_ + 1 * PB + sa0) = xb2;                                    \
    *reinterpret_cast<u32x4_t*>(sb_ + 1 * PB + sa1) = xb3;                                    \
    *reinterpret_cast<u32x4_t*>(sb_ + 0 * PB + sb0) = xb4;                                    \
    *reinterpret_cast<u32x4_t*>(sb_ + 0 * PB + sb1) = xb5;                                    \
    *reinterpret_cast<u32x4_t*>(sb_ + 1 * PB + sb0) = xb6;                                    \
    *reinterpret_cast<u32x4_t*>(sb_ + 1 * PB + sb1) = xb7;                                    \
  }
#define WH2W_OCT(PA_, PB_)                                                                    \
  _Pragma("unroll") for (int i = 0; i < 2; i++)                                               \
    _Pragma("unroll") for (int j = 0; j < 4; j++)                                             \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[i][PA_], B_[j][PB_], acc[i][j], 0, 0, 0);

  WH2W_GLOAD()
  for (int it = 0; it < NK; it++) {
    WH2W_STORE()
    __syncthreads();
    WH2W_GLOAD()
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      f16x8_t A_[2][2], B_[4][2];
#pragma unroll
      for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int i = 0; i < 2; i++) A_[i][p] = *reinterpret_cast<const f16x8_t*>(lds + p * PA + h2_lds_off(ra[i], 2 * ks + kh));
#pragma unroll
        for (int j = 0; j < 4; j++) B_[j][p] = *reinterpret_cast<const f16x8_t*>(lds + 2 * PA + p * PB + h2_lds_off(rb[j], 2 * ks + kh));
      }
      WH2W_OCT(1, 0) WH2W_OCT(0, 1) WH2W_OCT(0, 0)
    }
    __syncthreads();
  }
#undef WH2W_OCT
#undef WH2W_STORE
#undef WH2W_GLOAD

#pragma unroll
  for (int i = 0; i < 2; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int m = m0 + row;
      if (m < a.T) {
        float* dst = a.Mb + h2_row(h, pos, m) * a.Ntot;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int c = n0 + wn * 128 + j * 32 + (lane & 31);
          if (c < a.Ntot) dst[c] = acc[i][j][r];
        }
      }
    }
  }
}

// Deep-prefetch GEMM for a compile-time K extent (NK 32-channel steps, fully unrolled), tile 128 x (128 * NT).
//  * The A operand — the only one that comes from HBM, every row a first touch — is fetched PFA steps ahead into rotating
//    register sets; B (L2-resident) one step ahead.  Every step issues B FIRST and A after it: vmcnt retires in order, so the
//    wait before the LDS stores is vmcnt(4) — B and the A set that is due have landed, the newest A set stays in flight
//    (issuing A first forces vmcnt(0) and defeats the prefetch: measured, PFA 1-4 within 5 %).
//  * Staging map: thread t owns the 16-byte chunk t%4 of rows t/4 + 64*k — eight consecutive lanes write 128 contiguous LDS
//    bytes (the plain kernels' (row t/2, half t%2) map makes every ds_write_b128 a 2-way bank conflict: SQ_LDS_BANK_CONFLICT
//    was 1/3 of SQ_LDS_IDX_ACTIVE), and four lanes read 64 contiguous global bytes.
//  * Full tiles store without per-element bounds branches.
template <int NK, int PFA, int NT>
__global__ __launch_bounds__(256, 2) void wino_gemm_h2d_kernel(WinoH2Args h) {
  const WinoArgs& a = h.w;
  constexpr int BN = 128 * NT;
  constexpr int PA = 128 * 64, PB = BN * 64;
  constexpr int STAGE = 2 * PA + 2 * PB;          // 32 KB (NT 1) / 48 KB (NT 2)
  constexpr int NJ = 2 * NT;                      // 32-column MFMA tiles per wave
  constexpr int NBR = 2 * NT;                     // B staging rows per thread
  __shared__ __attribute__((aligned(16))) unsigned char lds[STAGE];

  const int n_nt = (a.Ntot + BN - 1) / BN;
  const int per_pos = a.n_mtiles * n_nt;
  const int nblk = h.npos * per_pos;
  const int id = blockIdx.x;
  int q = nblk >> 3, rr = nblk & 7, xcd = id & 7, slot = id >> 3;
  int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + slot;
  const int pos = tile / per_pos;
  const int rem = tile - pos * per_pos;
  const int m_tile = rem / n_nt, n_tile = rem - m_tile * n_nt;
  const int m0 = m_tile * 128, n0 = n_tile * BN;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  const int sr = tid >> 2, sc = tid & 3;          // staging: rows sr + 64 k, 16-byte chunk sc
  const unsigned piece_bytes = (unsigned)a.Ntot * 64u;
  unsigned xo[2], so_a[2], wo[NBR], so_b[NBR];
#pragma unroll
  for (int k = 0; k < 2; k++) {
    int m = m0 + sr + 64 * k;
    if (m >= a.T) m = a.T - 1;
    xo[k] = (unsigned)((h2_row(h, pos, m) * a.C) * 4) + (unsigned)sc * 16u;
    so_a[k] = h2_lds_off(sr + 64 * k, sc);
  }
#pragma unroll
  for (int k = 0; k < NBR; k++) {
    int n = n0 + sr + 64 * k;
    if (n >= a.Ntot) n = a.Ntot - 1;
    wo[k] = (unsigned)pos * (unsigned)NK * 2u * piece_bytes + (unsigned)n * 64u + (unsigned)sc * 16u;
    so_b[k] = h2_lds_off(sr + 64 * k, sc);
  }

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < NJ; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  int ra[2], rb[NJ];
#pragma unroll
  for (int i = 0; i < 2; i++) ra[i] = (wm * 2 + i) * 32 + (lane & 31);
#pragma unroll
  for (int j = 0; j < NJ; j++) rb[j] = wn * (64 * NT) + j * 32 + (lane & 31);
  const int kh = lane >> 5;

  const char* xbase = reinterpret_cast<const char*>(a.V);
  const char* wbase = reinterpret_cast<const char*>(h.U2);
  u32x4_t xa[PFA][4];          // [set][row k * 2 + piece]
  u32x4_t xb[NBR][2];          // [row k][piece]
  auto load_a = [&](int set, int kk) {
#pragma unroll
    for (int k = 0; k < 2; k++) {
      xa[set][2 * k + 0] = *reinterpret_cast<const u32x4_t*>(xbase + xo[k] + (unsigned)kk * 128u);
      xa[set][2 * k + 1] = *reinterpret_cast<const u32x4_t*>(xbase + xo[k] + (unsigned)kk * 128u + 64u);
    }
  };
  auto load_b = [&](int kk) {
#pragma unroll
    for (int k = 0; k < NBR; k++) {
      xb[k][0] = *reinterpret_cast<const u32x4_t*>(wbase + wo[k] + (unsigned)kk * 2u * piece_bytes);
      xb[k][1] = *reinterpret_cast<const u32x4_t*>(wbase + wo[k] + (unsigned)kk * 2u * piece_bytes + piece_bytes);
    }
  };
  load_b(0);
#pragma unroll
  for (int p = 0; p < PFA; p++) load_a(p, p < NK ? p : NK - 1);
#pragma unroll
  for (int it = 0; it < NK; it++) {
    const int set = it % PFA;
#pragma unroll
    for (int k = 0; k < 2; k++) {
      *reinterpret_cast<u32x4_t*>(lds + 0 * PA + so_a[k]) = xa[set][2 * k + 0];
      *reinterpret_cast<u32x4_t*>(lds + 1 * PA + so_a[k]) = xa[set][2 * k + 1];
    }
#pragma unroll
    for (int k = 0; k < NBR; k++) {
      *reinterpret_cast<u32x4_t*>(lds + 2 * PA + 0 * PB + so_b[k]) = xb[k][0];
      *reinterpret_cast<u32x4_t*>(lds + 2 * PA + 1 * PB + so_b[k]) = xb[k][1];
    }
    __syncthreads();
    if (it + 1 < NK) load_b(it + 1);            // B first: see the header
    if (it + PFA < NK) load_a(set, it + PFA);
    // (no run-time conditions here: behind a run-time branch the compiler must assume the loads were NOT issued and waits with
    //  vmcnt(0) before the next stores — the measurement knob that switched these loads off did exactly that to every build)
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      f16x8_t A_[2][2], B_[NJ][2];
#pragma unroll
      for (int p = 0; p < 2; p++) {
#pragma unroll
        for (int i = 0; i < 2; i++) A_[i][p] = *reinterpret_cast<const f16x8_t*>(lds + p * PA + h2_lds_off(ra[i], 2 * ks + kh));
#pragma unroll
        for (int j = 0; j < NJ; j++) B_[j][p] = *reinterpret_cast<const f16x8_t*>(lds + 2 * PA + p * PB + h2_lds_off(rb[j], 2 * ks + kh));
      }
#pragma unroll
      for (int pp = 0; pp < 3; pp++) {            // small terms first: lo*hi, hi*lo, hi*hi
        const int pa = pp == 0 ? 1 : 0, pb = pp == 1 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < NJ; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[i][pa], B_[j][pb], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  const bool full = m0 + 128 <= a.T && n0 + BN <= a.Ntot;   // uniform
  float* dst0 = a.Mb + h2_row(h, pos, m0 + wm * 64 + 4 * (lane >> 5)) * a.Ntot + n0 + wn * (64 * NT) + (lane & 31);
  if (full) {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        float* d = dst0 + (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * a.Ntot;
#pragma unroll
        for (int j = 0; j < NJ; j++) d[j * 32] = acc[i][j][r];
      }
  } else {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const int m = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < a.T) {
          float* d = a.Mb + h2_row(h, pos, m) * a.Ntot;
#pragma unroll
          for (int j = 0; j < NJ; j++) {
            const int c = n0 + wn * (64 * NT) + j * 32 + (lane & 31);
            if (c < a.Ntot) d[c] = acc[i][j][r];
          }
        }
      }
  }
}

// (Measured and dropped, profiles/r02/gemm_double_buffer_ab.log: a double-buffered LDS stage with ONE barrier per K step and the next
// step's stores issued between the two MFMA groups — 64 KB per workgroup, two workgroups per CU — 0.445 ms against 0.409 for the
// single-stage 128x128 form at four workgroups per CU and 0.389 for the 128x256 form: occupancy beats barrier count here.)

// wino_out_kernel + exact un-scaling + the per-board maximum of the block output (the next block's range).
template <int TM>
__global__ __launch_bounds__(256) void wino_out_h2_kernel(WinoH2Args h) {
  using WT = WinoT<TM>;
  constexpr int AL = WT::AL;
  const WinoArgs& a = h.w;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = g < (size_t)a.T * a.Cout_p;
  const size_t gg = live ? g : (size_t)a.T * a.Cout_p - 1;
  const int c = (int)(gg % a.Cout_p);
  const int t = (int)(gg / a.Cout_p);
  const int b = t / a.TPB, tt = t - b * a.TPB;
  const int ty = tt / a.ntx, tx = tt - ty * a.ntx;
  float s_, unscale;
  wino_h2_scales(h.amax_in[b], WT::VSHIFT, &s_, &unscale);
  const float un_a = unscale * (h.col_unscale ? h.col_unscale[c] : h.w_unscale);
  const float un_b = unscale * (h.col_unscale ? h.col_unscale[a.Cout_p + c] : h.w_unscale);
  const float tn = h.t_next ? h.t_next[c] : 1.f;
  // All loads are issued before any arithmetic: the 2 x AL^2 values of M and the TM^2 epilogue parameter vectors (clamped
  // address) — loaded inside the per-pixel branch each of those is a dependent L2 round trip per thread.  (Fetching branch by
  // branch to lower the register count was measured: the compiler hoists the loads anyway and spills — 0.51 vs 0.24 ms.)
  float4 E[TM][TM];
  {
    const float4* ep = reinterpret_cast<const float4*>(a.ep);
#pragma unroll
    for (int k = 0; k < TM; k++) {
      const int hh = TM * ty + k, hc = hh < a.H ? hh : a.H - 1;
#pragma unroll
      for (int l = 0; l < TM; l++) {
        const int ww = TM * tx + l, wc = ww < a.W ? ww : a.W - 1;
        E[k][l] = ep[(size_t)(hc * a.W + wc) * a.Cout_p + c];
      }
    }
  }
  float Y[2][TM][TM];
#pragma unroll
  for (int br = 0; br < 2; br++) {
    float tm_[TM][AL];
#pragma unroll
    for (int nu = 0; nu < AL; nu++) {
      float m[AL], o[TM];
#pragma unroll
      for (int xi = 0; xi < AL; xi++) m[xi] = a.Mb[h2_row(h, xi * AL + nu, t) * a.Ntot + br * a.Cout_p + c];
      wino_atv<TM>(m, o);
#pragma unroll
      for (int k = 0; k < TM; k++) tm_[k][nu] = o[k];
    }
#pragma unroll
    for (int k = 0; k < TM; k++) wino_atv<TM>(tm_[k], Y[br][k]);
  }
  float* yb = a.y + (size_t)b * a.Hp * a.Wp * a.Cout_p + c;
  float mx = 0.f;
#pragma unroll
  for (int k = 0; k < TM; k++) {
    const int hh = TM * ty + k;
#pragma unroll
    for (int l = 0; l < TM; l++) {
      const int ww = TM * tx + l;
      const float4 e = E[k][l];
      float va = (Y[0][k][l] * un_a) * e.x + e.y;
      float vb = (Y[1][k][l] * un_b) * e.z + e.w;
      va = va > 0.f ? va : 0.f;
      vb = vb > 0.f ? vb : 0.f;
      float s = va + vb;
      s = s > 0.f ? s : 0.f;
      if (live && hh < a.H && ww < a.W) {
        yb[((size_t)(hh + 1) * a.Wp + (ww + 1)) * a.Cout_p] = s;
        mx = fmaxf(mx, s * tn);
      }
    }
  }
  // the 64 lanes of a wave are 64 channels of ONE tile (Cout_p is a multiple of 64): one plain store per wave, reduced per board by
  // wino_board_max_kernel (an atomicMax per wave into the board's word cost 0.03 of the kernel's 0.25 ms: scripts/probes/stream_probe P7)
  if (h.wave_max) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0 && live) h.wave_max[(size_t)t * (a.Cout_p >> 6) + (c >> 6)] = mx;
  }
}

// ---- block-per-tile transform kernels -----------------------------------------------------------------------------------
// What the measurements of round 2 said about the thread-per-(tile, channel) kernel above on the F(5x5,3x3) shape
// (profiles/r02/wino_h2_out_forms_ab.log, wino_out_decomposition.log, pmc_wino_out_tile_form.json):
//  * ~1100 vector instructions per thread, ~600 of them 64-bit address arithmetic (quarter-rate v_mul_lo_u32 / v_mad_u64_u32 for
//    every one of 74 loads).  With ONE tile per workgroup, tile / board / pixel / position are uniform and every address is a
//    buffer descriptor over a scalar base + one per-lane 32-bit offset + a scalar offset: 350 vector instructions — and no time
//    gained (0.282 -> 0.277 ms): instruction issue was not the limit.
//  * A lane-pair mapping (lanes 2i, 2i+1 = branches a, b of one channel; two 128-byte runs per memory instruction) measured
//    0.31 ms where one 256-byte run per instruction measured 0.24 ms for the same bytes (F(4x4) shape): the memory pipeline's
//    throughput goes with the bytes one instruction moves in one run.  Both lane-pair kernels were removed after the A/B.
// Kept: scalar buffer addressing, packed-fp32 1-D transforms two columns / rows at a time, and the kernel below.
typedef float f2v __attribute__((ext_vector_type(2)));
// Buffer addressing for the block-per-tile transform kernels: a scalar descriptor over a block-uniform base, one per-lane 32-bit
// byte offset and a scalar byte offset per access (position / pixel) — no vector address arithmetic at all.  (Plain pointer
// arithmetic does not get there: the compiler re-associates base + lane offset first and then adds every position's offset
// with a 64-bit v_mad_u64_u32 per load.)  Raw buffer, range check on the lane offset only.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t h2_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ float h2_ldf(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
// 8-byte load.  NOTE: `__builtin_bit_cast(T, vec[i])` applied directly to a vector ELEMENT expression reads element 0 with
// this compiler (seen in the ISA: a b64 load narrowed to one dword, both halves of a permlane swap storing the same register);
// elements are copied to scalars first.
__device__ __forceinline__ float2 h2_ldf2(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
  const unsigned x = v[0], y = v[1];
  return make_float2(__uint_as_float(x), __uint_as_float(y));
}
template <typename T> __device__ __forceinline__ T h2_ldg(const void* sbase, unsigned voff_bytes) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(sbase) + voff_bytes);
}
__device__ __forceinline__ void h2_stf(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}
template <int TM, typename V> __device__ __forceinline__ void wino_atv_t(const V* m, V* o);
template <> __device__ __forceinline__ void wino_atv_t<4, float>(const float* m, float* o) { wino_at4(m, o); }
template <> __device__ __forceinline__ void wino_atv_t<5, float>(const float* m, float* o) { wino_atv<5>(m, o); }
template <> __device__ __forceinline__ void wino_atv_t<4, f2v>(const f2v* m, f2v* o) {
  const f2v s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  o[0] = m[0] + s12 + s34;
  o[1] = d12 + 2.f * d34;
  o[2] = s12 + 4.f * s34;
  o[3] = d12 + 8.f * d34 + m[5];
}
template <> __device__ __forceinline__ void wino_atv_t<5, f2v>(const f2v* m, f2v* o) {
  const f2v s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  o[0] = m[0] + s12 + s34 + m[5];
  o[1] = d12 + 0.5f * d34 + 2.f * m[5];
  o[2] = s12 + 0.25f * s34 + 4.f * m[5];
  o[3] = d12 + 0.125f * d34 + 8.f * m[5];
  o[4] = s12 + 0.0625f * s34 + 16.f * m[5] + m[6];
}

// Block-per-tile form with one thread per channel and the two branches one after the other: a wave's 64 lanes are 64 consecutive
// channels, so every M load and every y store of a wave is one 256-byte run, and finishing branch a before
// branch b's loads are issued keeps the live set at ~49 + 50 + 25 values.  The asm barrier keeps the compiler from hoisting
// branch b's loads to the top (which is what made the thread-per-channel kernel above need 256 registers).  (Also tried: the
// parameter loads issued only after the first transform pass, 127 registers / 4 waves per SIMD: 0.249 against 0.238 ms; and,
// because the epilogue parameters come from L2 once per workgroup — as many bytes as the workgroup's M — a (tile position,
// channel group)-major workgroup order with 64 / 128 / 256 threads so that a CU's resident workgroups share their parameter
// slice: 0.251 / 0.273 / 0.260 against 0.236 ms, profiles/r02/wino_out_workgroup_order_ab.log.)
template <int TM>
__global__ __launch_bounds__(256) void wino_out_seq_h2_kernel(WinoH2Args h) {
  using WT = WinoT<TM>;
  constexpr int AL = WT::AL;
  const WinoArgs& a = h.w;
  const int t = blockIdx.x;                                        // uniform: one tile per block
  const int tid = threadIdx.x, c = blockIdx.y * blockDim.x + tid;  // this lane's channel
  const int b = t / a.TPB, tt = t - b * a.TPB;
  const int ty = tt / a.ntx, tx = tt - ty * a.ntx;
  float s_, unscale0;
  wino_h2_scales(h.amax_in[b], WT::VSHIFT, &s_, &unscale0);
  const float tn = h.t_next ? h.t_next[c] : 1.f;
  const __amdgpu_buffer_rsrc_t mr = h2_rsrc(a.Mb + ((size_t)(t >> h.rsh) * h.rA + (size_t)(t & h.rmask)) * a.Ntot);
  const unsigned pos_stride = h.rB * (unsigned)a.Ntot * 4u;        // bytes, uniform
  const __amdgpu_buffer_rsrc_t er = h2_rsrc(a.ep);                 // float2 index (pixel * Cout_p + c) * 2 + branch
  const unsigned e_pix = (unsigned)a.Cout_p * 16u;                 // bytes per pixel
  float va[TM][TM];
#pragma unroll
  for (int br = 0; br < 2; br++) {
    const unsigned lane_off = (unsigned)(br * a.Cout_p + c) * 4u;
    const unsigned e_lane = (unsigned)(c * 2 + br) * 8u;
    const float unscale = unscale0 * (h.col_unscale ? h.col_unscale[br * a.Cout_p + c] : h.w_unscale);
    float tm_[TM][AL];
#pragma unroll
    for (int nu = 0; nu + 1 < AL; nu += 2) {       // two columns at a time on the packed-fp32 pipe
      f2v m[AL], o[TM];
#pragma unroll
      for (int xi = 0; xi < AL; xi++) {
        m[xi].x = h2_ldf(mr, lane_off, (unsigned)(xi * AL + nu) * pos_stride);
        m[xi].y = h2_ldf(mr, lane_off, (unsigned)(xi * AL + nu + 1) * pos_stride);
      }
      wino_atv_t<TM, f2v>(m, o);
#pragma unroll
      for (int k = 0; k < TM; k++) { tm_[k][nu] = o[k].x; tm_[k][nu + 1] = o[k].y; }
    }
    if (AL & 1) {
      constexpr int nu = AL - 1;
      float m[AL], o[TM];
#pragma unroll
      for (int xi = 0; xi < AL; xi++) m[xi] = h2_ldf(mr, lane_off, (unsigned)(xi * AL + nu) * pos_stride);
      wino_atv_t<TM, float>(m, o);
#pragma unroll
      for (int k = 0; k < TM; k++) tm_[k][nu] = o[k];
    }
    float2 E[TM][TM];
#pragma unroll
    for (int k = 0; k < TM; k++) {
      const int hh = TM * ty + k, hc = hh < a.H ? hh : a.H - 1;
#pragma unroll
      for (int l = 0; l < TM; l++) {
        const int ww = TM * tx + l, wc = ww < a.W ? ww : a.W - 1;
        E[k][l] = h2_ldf2(er, e_lane, (unsigned)(hc * a.W + wc) * e_pix);
      }
    }
    float Y[TM][TM];
#pragma unroll
    for (int k = 0; k + 1 < TM; k += 2) {          // two rows at a time
      f2v r[AL], o[TM];
#pragma unroll
      for (int nu = 0; nu < AL; nu++) { r[nu].x = tm_[k][nu]; r[nu].y = tm_[k + 1][nu]; }
      wino_atv_t<TM, f2v>(r, o);
#pragma unroll
      for (int l = 0; l < TM; l++) { Y[k][l] = o[l].x; Y[k + 1][l] = o[l].y; }
    }
    if (TM & 1) wino_atv_t<TM, float>(tm_[TM - 1], Y[TM - 1]);
#pragma unroll
    for (int k = 0; k < TM; k++)
#pragma unroll
      for (int l = 0; l < TM; l++) {
        float v = (Y[k][l] * unscale) * E[k][l].x + E[k][l].y;
        v = v > 0.f ? v : 0.f;
        va[k][l] = br == 0 ? v : va[k][l] + v;      // relu(a) + relu(b) >= 0 already
      }
    if (br == 0) {   // branch a's results exist before any of branch b's loads is issued (values through an ordered empty asm)
#pragma unroll
      for (int k = 0; k < TM; k++)
#pragma unroll
        for (int l = 0; l < TM; l++) asm volatile("" : "+v"(va[k][l]) : : "memory");
    }
  }
  const __amdgpu_buffer_rsrc_t yr = h2_rsrc(a.y + (size_t)b * a.Hp * a.Wp * a.Cout_p);
  const unsigned y_lane = (unsigned)c * 4u, y_pix = (unsigned)a.Cout_p * 4u;
  float mx = 0.f;
#pragma unroll
  for (int k = 0; k < TM; k++) {
    const int hh = TM * ty + k;
#pragma unroll
    for (int l = 0; l < TM; l++) {
      const int ww = TM * tx + l;
      if (hh < a.H && ww < a.W) {                   // uniform
        h2_stf(yr, y_lane, (unsigned)((hh + 1) * a.Wp + (ww + 1)) * y_pix, va[k][l]);
        mx = fmaxf(mx, va[k][l] * tn);
      }
    }
  }
  if (h.wave_max) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) h.wave_max[(size_t)t * (a.Cout_p >> 6) + (c >> 6)] = mx;
  }
}

// Quad form of the same transform (Cout_p a multiple of 256): every global access of a wave is ONE 1 KB run.  The memory pipeline's
// throughput on this part goes with the bytes an instruction moves in one run (128-byte runs 0.31 ms, 256-byte runs 0.24 ms for the
// same bytes, profiles/r02; a bare float4 stream with this kernel's read/write mix: 0.205 ms, profiles/r03/rw_probe.log), and one
// thread per channel cannot do better than 256 bytes.  Here a lane owns FOUR consecutive channels and the 2-D transform is split
// between waves through LDS: phase 1, wave nu = column nu of the position grid: AL float4 loads of M per branch (the column), the
// column pass At . on each of the four channels, TM float4 per branch into LDS; phase 2, wave k = output row k: the row pass over
// the AL columns from LDS, un-scale, the block epilogue on float4 parameters, TM float4 stores of y.  Same arithmetic in the same
// order as the thread-per-channel kernel (bit-identical results, checked on a 300-board K=256 network).  Measured 0.251 -> 0.215 ms on
// the headline block = 4.7 TB/s, 0.95 of the stream ceiling of its mix.  LDS: 72 KB per tile for F(5x5,3x3), two workgroups per CU.
template <int AUX = 0>   // AUX = 2: non-temporal (a stream read once)
__device__ __forceinline__ float4 h2_ldf4(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, AUX);
  const unsigned x = v[0], y = v[1], z = v[2], w = v[3];
  return make_float4(__uint_as_float(x), __uint_as_float(y), __uint_as_float(z), __uint_as_float(w));
}
template <int TM>
__global__ __launch_bounds__((TM + 2) * 64) void wino_out_quad_h2_kernel(WinoH2Args h) {
  using WT = WinoT<TM>;
  constexpr int AL = WT::AL;
  const WinoArgs& a = h.w;
  __shared__ __attribute__((aligned(16))) float S[2][TM][AL][256];   // column-pass results [branch][row k][column nu][channel]
  __shared__ float wmx[AL];
  const int t = blockIdx.x;                                          // one tile per workgroup
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c0 = blockIdx.y * 256 + 4 * lane;                        // this lane's four channels c0 .. c0 + 3
  const bool act = c0 < a.Cout_p;                                    // (Cout_p = 128: half of every wave has no channels)
  const int b = t / a.TPB, tt = t - b * a.TPB;
  const int ty = tt / a.ntx, tx = tt - ty * a.ntx;
  float s_, unscale0;
  wino_h2_scales(h.amax_in[b], WT::VSHIFT, &s_, &unscale0);
  // ---- phase 1: column nu = wv of the AL x AL position grid, both branches
  if (act) {
    const int nu = wv;
    const __amdgpu_buffer_rsrc_t mr = h2_rsrc(a.Mb + ((size_t)(t >> h.rsh) * h.rA + (size_t)(t & h.rmask)) * a.Ntot);
    const unsigned pos_stride = h.rB * (unsigned)a.Ntot * 4u;        // bytes, uniform
    float4 m[2][AL];
#pragma unroll
    for (int br = 0; br < 2; br++) {
      const unsigned lane_off = (unsigned)(br * a.Cout_p + c0) * 4u;
#pragma unroll
      for (int xi = 0; xi < AL; xi++) m[br][xi] = h2_ldf4(mr, lane_off, (unsigned)(xi * AL + nu) * pos_stride);
    }
#pragma unroll
    for (int br = 0; br < 2; br++) {
      float mm[4][AL], oo[4][TM];
#pragma unroll
      for (int xi = 0; xi < AL; xi++) { mm[0][xi] = m[br][xi].x; mm[1][xi] = m[br][xi].y; mm[2][xi] = m[br][xi].z; mm[3][xi] = m[br][xi].w; }
#pragma unroll
      for (int e = 0; e < 4; e++) wino_atv<TM>(mm[e], oo[e]);
#pragma unroll
      for (int k = 0; k < TM; k++) *reinterpret_cast<float4*>(&S[br][k][nu][4 * lane]) = make_float4(oo[0][k], oo[1][k], oo[2][k], oo[3][k]);
    }
  }
  __syncthreads();
  // ---- phase 2: output row k = wv (waves TM .. AL-1 have no row)
  float mx = 0.f;
  if (wv < TM && act) {
    const int k = wv;
    const int hh = TM * ty + k;                                      // uniform
    float Y[2][4][TM];
#pragma unroll
    for (int br = 0; br < 2; br++) {
      float rr[4][AL];
#pragma unroll
      for (int nu = 0; nu < AL; nu++) {
        const float4 v = *reinterpret_cast<const float4*>(&S[br][k][nu][4 * lane]);
        rr[0][nu] = v.x; rr[1][nu] = v.y; rr[2][nu] = v.z; rr[3][nu] = v.w;
      }
#pragma unroll
      for (int e = 0; e < 4; e++) wino_atv<TM>(rr[e], Y[br][e]);
    }
    float ua[4], ub[4], tn[4];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      ua[e] = unscale0 * (h.col_unscale ? h.col_unscale[c0 + e] : h.w_unscale);
      ub[e] = unscale0 * (h.col_unscale ? h.col_unscale[a.Cout_p + c0 + e] : h.w_unscale);
      tn[e] = h.t_next ? h.t_next[c0 + e] : 1.f;
    }
    if (hh < a.H) {
      const __amdgpu_buffer_rsrc_t er = h2_rsrc(a.ep);               // float4 {sa, ta, sb, tb} per (pixel, channel)
      const __amdgpu_buffer_rsrc_t yr = h2_rsrc(a.y + (size_t)b * a.Hp * a.Wp * a.Cout_p);
      const unsigned e_lane = (unsigned)c0 * 16u, e_pix = (unsigned)a.Cout_p * 16u;
      const unsigned y_lane = (unsigned)c0 * 4u, y_pix = (unsigned)a.Cout_p * 4u;
#pragma unroll
      for (int l = 0; l < TM; l++) {
        const int ww = TM * tx + l;
        if (ww < a.W) {                                              // uniform
          const unsigned pe = (unsigned)(hh * a.W + ww) * e_pix;
          float4 E[4];
#pragma unroll
          for (int e = 0; e < 4; e++) E[e] = h2_ldf4(er, e_lane + 16u * e, pe);
          float o[4];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float va = (Y[0][e][l] * ua[e]) * E[e].x + E[e].y;
            float vb = (Y[1][e][l] * ub[e]) * E[e].z + E[e].w;
            va = va > 0.f ? va : 0.f;
            vb = vb > 0.f ? vb : 0.f;
            o[e] = va + vb;                                          // relu(a) + relu(b) >= 0 already
            mx = fmaxf(mx, o[e] * tn[e]);
          }
          const unsigned ubits[4] = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
          typedef unsigned u32x4_o __attribute__((ext_vector_type(4)));
          __builtin_amdgcn_raw_buffer_store_b128(u32x4_o{ubits[0], ubits[1], ubits[2], ubits[3]}, yr, y_lane, (unsigned)((hh + 1) * a.Wp + (ww + 1)) * y_pix, 0);
        }
      }
    }
  }
  if (h.wave_max) {   // the tile's maximum, written to each of its 64-channel words (the consumer takes the board's maximum of all words)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) wmx[wv] = mx;
    __syncthreads();
    const int n_words = min(4, (a.Cout_p - (int)blockIdx.y * 256) >> 6);
    if (tid < n_words) {
      float m2 = wmx[0];
#pragma unroll
      for (int u = 1; u < AL; u++) m2 = fmaxf(m2, wmx[u]);
      h.wave_max[(size_t)t * (a.Cout_p >> 6) + blockIdx.y * 4 + tid] = m2;
    }
  }
}

// Output transform without an epilogue (dual.Train's forward and data-gradient convolutions, train.hip): y = At M A * (1 / (s_b su)),
// every one of the Ntot GEMM columns its own output channel.  One workgroup per tile, one thread per column, buffer addressing as above.
template <int TM>
__global__ __launch_bounds__(256) void wino_out_raw_h2_kernel(WinoH2Args h) {
  using WT = WinoT<TM>;
  constexpr int AL = WT::AL;
  const WinoArgs& a = h.w;
  const int t = blockIdx.x;                                        // uniform: one tile per block
  const int tid = threadIdx.x, c = blockIdx.y * blockDim.x + tid;  // this lane's column
  const int b = t / a.TPB, tt = t - b * a.TPB;
  const int ty = tt / a.ntx, tx = tt - ty * a.ntx;
  float s_, unscale;
  wino_h2_scales(h.amax_in[b], WT::VSHIFT, &s_, &unscale);
  unscale *= h.w_unscale_dev[0];
  const __amdgpu_buffer_rsrc_t mr = h2_rsrc(a.Mb + ((size_t)(t >> h.rsh) * h.rA + (size_t)(t & h.rmask)) * a.Ntot);
  const unsigned pos_stride = h.rB * (unsigned)a.Ntot * 4u;
  const unsigned lane_off = (unsigned)c * 4u;
  float tm_[TM][AL];
#pragma unroll
  for (int nu = 0; nu + 1 < AL; nu += 2) {
    f2v m[AL], o[TM];
#pragma unroll
    for (int xi = 0; xi < AL; xi++) {
      m[xi].x = h2_ldf(mr, lane_off, (unsigned)(xi * AL + nu) * pos_stride);
      m[xi].y = h2_ldf(mr, lane_off, (unsigned)(xi * AL + nu + 1) * pos_stride);
    }
    wino_atv_t<TM, f2v>(m, o);
#pragma unroll
    for (int k = 0; k < TM; k++) { tm_[k][nu] = o[k].x; tm_[k][nu + 1] = o[k].y; }
  }
  if (AL & 1) {
    constexpr int nu = AL - 1;
    float m[AL], o[TM];
#pragma unroll
    for (int xi = 0; xi < AL; xi++) m[xi] = h2_ldf(mr, lane_off, (unsigned)(xi * AL + nu) * pos_stride);
    wino_atv_t<TM, float>(m, o);
#pragma unroll
    for (int k = 0; k < TM; k++) tm_[k][nu] = o[k];
  }
  const __amdgpu_buffer_rsrc_t yr = h2_rsrc(a.y + (size_t)b * a.Hp * a.Wp * a.Ntot);
  const unsigned y_pix = (unsigned)a.Ntot * 4u;
#pragma unroll
  for (int k = 0; k < TM; k++) {
    float Yk[TM];
    wino_atv_t<TM, float>(tm_[k], Yk);
    const int hh = TM * ty + k;
#pragma unroll
    for (int l = 0; l < TM; l++) {
      const int ww = TM * tx + l;
      if (hh < a.H && ww < a.W) h2_stf(yr, lane_off, (unsigned)((hh + 1) * a.Wp + (ww + 1)) * y_pix, Yk[l] * unscale);   // uniform condition
    }
  }
}

// Winograd-domain weights built ON THE DEVICE (training: the filters change every step).  w [tap 9][N][C] fp32 ->
// u2[pos][C/32][piece 2][N][32] fp16 with the layer's power-of-two scale (max|U| su in [2^13, 2^14), like wino_build_u2 on the
// host); pass 1 reduces max|U| into umax_bits, pass 2 scales, splits and stores and writes 1 / su.  One thread per (n, ci).
template <int TM>
__device__ __forceinline__ void wino_u_of(const float* __restrict__ w, int N, int C, int n, int ci, float* U) {
  using WT = WinoT<TM>;
  constexpr int AL = WT::AL;
  double g[3][3], tg[AL][3];
#pragma unroll
  for (int tap = 0; tap < 9; tap++) g[tap / 3][tap % 3] = (double)w[((size_t)tap * N + n) * C + ci];
#pragma unroll
  for (int xi = 0; xi < AL; xi++)
#pragma unroll
    for (int j = 0; j < 3; j++) tg[xi][j] = WT::G[xi][0] * g[0][j] + WT::G[xi][1] * g[1][j] + WT::G[xi][2] * g[2][j];
#pragma unroll
  for (int xi = 0; xi < AL; xi++)
#pragma unroll
    for (int nu = 0; nu < AL; nu++)
      U[xi * AL + nu] = (float)(tg[xi][0] * WT::G[nu][0] + tg[xi][1] * WT::G[nu][1] + tg[xi][2] * WT::G[nu][2]);
}
template <int TM>
__global__ __launch_bounds__(256) void wino_u_absmax_kernel(const float* __restrict__ w, int N, int C, unsigned* __restrict__ umax_bits) {
  constexpr int NP = WinoT<TM>::AL * WinoT<TM>::AL;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  float mx = 0.f;
  if (g < (size_t)N * C) {
    float U[NP];
    wino_u_of<TM>(w, N, C, (int)(g / C), (int)(g % C), U);
#pragma unroll
    for (int i = 0; i < NP; i++) mx = fmaxf(mx, fabsf(U[i]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(umax_bits, __float_as_uint(mx));
}
template <int TM>
__global__ __launch_bounds__(256) void wino_u_build_kernel(const float* __restrict__ w, int N, int C, const unsigned* __restrict__ umax_bits,
                                                           _Float16* __restrict__ u2, float* __restrict__ unscale_out) {
  constexpr int NP = WinoT<TM>::AL * WinoT<TM>::AL;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  float su, inv;
  h2_scales(umax_bits[0], &su, &inv);      // 2^(13 - floor(log2 max|U|)) and its inverse
  if (g == 0) unscale_out[0] = inv;
  if (g >= (size_t)N * C) return;
  const int n = (int)(g / C), ci = (int)(g % C);
  float U[NP];
  wino_u_of<TM>(w, N, C, n, ci, U);
  const int NC = C / 32;
#pragma unroll
  for (int pos = 0; pos < NP; pos++) {
    const float xs = U[pos] * su;
    const _Float16 hi = (_Float16)xs;
    const _Float16 lo = (_Float16)(xs - (float)hi);
    const size_t base = ((((size_t)pos * NC + ci / 32) * 2) * N + n) * 32 + (ci % 32);
    u2[base] = hi;
    u2[base + (size_t)N * 32] = lo;
  }
}

// amax_out[b] = max over the board's tiles and channel groups of wave_max (one wave per board)
__global__ __launch_bounds__(64) void wino_board_max_kernel(const float* __restrict__ wave_max, unsigned* __restrict__ amax_out, int per_board) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float mx = 0.f;
  for (int i = lane; i < per_board; i += 64) mx = fmaxf(mx, wave_max[(size_t)b * per_board + i]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (lane == 0) amax_out[b] = __float_as_uint(mx);
}

// Host: the Winograd-domain filter G g Gt (double, rounded once to fp32), EQUILIBRATED by powers of two — input channel ci divided by
// t_in[ci] = 2^floor(log2 max_{pos,n} |U|) (the input transform multiplies V's channel ci by the same factor: exact, the products do
// not change), then every column n scaled by its own su_n with max_{pos,ci} |U / t| * su_n in [2^13, 2^14) — and split into two fp16
// pieces: u2[pos][ci/32][piece][n][ci%32].  col_unscale[n] = 1 / su_n.  A layer whose filters or input channels differ by many
// octaves keeps 22-bit pieces in every row and column that matters (tests/test_wino_gpu.py, heterogeneous ranges).
template <int TM, typename Get>
static void wino_build_u2(std::vector<_Float16>& u2, int Ntot, int C, Get get, std::vector<float>& t_in, std::vector<float>& col_unscale) {
  using WT = WinoT<TM>;
  constexpr int AL = WT::AL, NP = AL * AL;
  const int NC = C / 32;
  std::vector<float> U((size_t)NP * Ntot * C, 0.f);
  std::vector<float> rmax(C, 0.f);
  for (int n = 0; n < Ntot; n++)
    for (int ci = 0; ci < C; ci++) {
      double g[3][3], tg[AL][3];
      bool any = false;
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { g[i][j] = get(n, ci, i * 3 + j); any = any || g[i][j] != 0.0; }
      if (!any) continue;
      for (int xi = 0; xi < AL; xi++) for (int j = 0; j < 3; j++) tg[xi][j] = WT::G[xi][0] * g[0][j] + WT::G[xi][1] * g[1][j] + WT::G[xi][2] * g[2][j];
      for (int xi = 0; xi < AL; xi++) for (int nu = 0; nu < AL; nu++) {
        const float v = (float)(tg[xi][0] * WT::G[nu][0] + tg[xi][1] * WT::G[nu][1] + tg[xi][2] * WT::G[nu][2]);
        U[((size_t)(xi * AL + nu) * Ntot + n) * C + ci] = v;
        rmax[ci] = std::max(rmax[ci], std::fabs(v));
      }
    }
  t_in.assign(C, 1.0f);
  for (int ci = 0; ci < C; ci++)
    if (rmax[ci] > 0.f && std::isfinite(rmax[ci])) {
      int ex = 0;
      std::frexp(rmax[ci], &ex);                       // rmax = m * 2^ex, m in [0.5, 1)
      ex = std::max(-100, std::min(100, ex - 1));      // (clamped: t and 1/t stay normal fp32 numbers with room for the activations)
      t_in[ci] = std::ldexp(1.0f, ex);
    }
  std::vector<float> cmax(Ntot, 0.f);
  for (int pos = 0; pos < NP; pos++)
    for (int n = 0; n < Ntot; n++)
      for (int ci = 0; ci < C; ci++) {
        float& v = U[((size_t)pos * Ntot + n) * C + ci];
        v /= t_in[ci];                                  // exact (power of two)
        cmax[n] = std::max(cmax[n], std::fabs(v));
      }
  col_unscale.assign(Ntot, 1.0f);
  std::vector<float> su(Ntot, 1.0f);
  for (int n = 0; n < Ntot; n++)
    if (cmax[n] > 0.f && std::isfinite(cmax[n])) {
      int ex = 0;
      std::frexp(cmax[n], &ex);
      su[n] = std::ldexp(1.0f, 14 - ex);
      col_unscale[n] = 1.0f / su[n];
    }
  u2.assign((size_t)NP * NC * 2 * Ntot * 32, (_Float16)0.f);
  for (int pos = 0; pos < NP; pos++)
    for (int n = 0; n < Ntot; n++)
      for (int ci = 0; ci < C; ci++) {
        const float xs = U[((size_t)pos * Ntot + n) * C + ci] * su[n];
        const _Float16 hi = (_Float16)xs;
        const _Float16 lo = (_Float16)(xs - (float)hi);
        const size_t base = ((((size_t)pos * NC + ci / 32) * 2) * Ntot + n) * 32 + (ci % 32);
        u2[base] = hi;
        u2[base + (size_t)Ntot * 32] = lo;
      }
}

// tile size with the fewest transform-domain rows for an H x W board
static inline int wino_h2_pick_tm(int H, int W) {
  static const int force = [] { const char* e = getenv("AGZ_WINO_H2_TM"); return e ? atoi(e) : 0; }();   // environment switch (agz.h)
  if (force == 4 || force == 5) return force;
  const int r4 = 36 * ceil_div(H, 4) * ceil_div(W, 4), r5 = 49 * ceil_div(H, 5) * ceil_div(W, 5);
  return r5 < r4 ? 5 : 4;
}

// rows of V (and of M) the launch below addresses for `tiles` tiles: what the caller sizes the buffers by
static inline size_t wino_h2_rows(int npos, size_t tiles, int row_pad = 0) { return (size_t)npos * (((tiles + 127) / 128) * (size_t)(128 + row_pad)); }

// launches the stages of one block for a chunk of boards (h.w.V / Mb sized by the caller: wino_h2_rows())
static void wino_h2_launch(agz_ctx* ctx, WinoH2Args& h, bool wide, hipStream_t st = nullptr) {
  if (!st) st = ctx->stream;
  WinoArgs& a = h.w;
  const int tm = h.tm == 5 ? 5 : 4;
  h.tm = tm; h.npos = (tm + 2) * (tm + 2);
  a.nty = ceil_div(a.H, tm); a.ntx = ceil_div(a.W, tm); a.TPB = a.nty * a.ntx; a.T = a.B * a.TPB;
  h.rsh = 7; h.rmask = 127; h.rB = 128u + (unsigned)h.row_pad; h.rA = (unsigned)h.npos * h.rB;   // V and M in blocks of 128 tiles: [tile / 128][position][tile % 128 (+ pad)]
  a.n_mtiles = ceil_div(a.T, 128); a.n_ntiles = ceil_div(a.Ntot, 128);
  h.in_swap = a.C % 128 == 0 ? 1 : 0;   // a wave = 64 channel pairs of ONE tile
  // Forms of the output transform: 0 = thread per (tile, channel), both branches at once;
  // 3 = block per tile, thread per channel, branch after branch.  Default: 3 for F(5x5,3x3) (0.237 ms against 0.30 for form 0 on
  // the headline block), 0 for F(4x4,3x3) (0.231 against 0.255 for form 3).  (1 and 2 were the removed lane-pair forms.)
  const bool fits32 = (size_t)h.npos * h.rB * a.Ntot * 4 < ((size_t)1 << 32);   // scalar position offsets of the tile form
  int form = tm == 5 ? 3 : 0;
  if (form == 3 && !fits32) form = 0;
  // the board-range reduction between two blocks rides in the next block's input transform
  const bool fuse = a.C % 128 == 0 && h.wave_max != nullptr;
  h.wm_per_board = a.TPB * (a.Cout_p >> 6);             // wave_max: one word per tile and 64 channels
  h.fuse_prev = (h.fuse_prev && fuse) ? 1 : 0;
  h.amax_self = const_cast<unsigned*>(h.amax_in);
  {
    ProfScopeOn ps(ctx, AGZ_PROF_WINO_IN, st == ctx->stream);
    const size_t n_in = (size_t)a.T * (a.C / 2);
    if (tm == 5) hipLaunchKernelGGL(wino_in_h2_kernel<5>, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, st, h);
    else hipLaunchKernelGGL(wino_in_h2_kernel<4>, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, st, h);
  }
  {
    ProfScopeOn ps(ctx, AGZ_PROF_WINO_GEMM, st == ctx->stream);
    const dim3 gw(h.npos * a.n_mtiles * ceil_div(a.Ntot, 256)), gn(h.npos * a.n_mtiles * a.n_ntiles);
    // the unrolled form with the A operand fetched two steps ahead, instantiated per K extent (32-channel steps); other extents take
    // the plain single-prefetch kernels
    const int nk = a.C >> 5;
    bool done = true;
#define AGZ_H2D(NK_, PF_, NT_, G_) hipLaunchKernelGGL((wino_gemm_h2d_kernel<NK_, PF_, NT_>), G_, dim3(256), 0, st, h)
#define AGZ_H2D_NK(NK_) { if (wide) AGZ_H2D(NK_, 2, 2, gw); else AGZ_H2D(NK_, 2, 1, gn); }
    if (a.C & 31) done = false;
    else if (nk == 8) AGZ_H2D_NK(8)
    else if (nk == 2) AGZ_H2D_NK(2)
    else if (nk == 4) AGZ_H2D_NK(4)
    else if (nk == 6) AGZ_H2D_NK(6)
    else if (nk == 12) AGZ_H2D_NK(12)
    else if (nk == 16) AGZ_H2D_NK(16)
    else done = false;
#undef AGZ_H2D_NK
#undef AGZ_H2D
    if (done) {}
    else if (wide) hipLaunchKernelGGL(wino_gemm_h2w_kernel, gw, dim3(256), 0, st, h);
    else hipLaunchKernelGGL(wino_gemm_h2_kernel, gn, dim3(256), 0, st, h);
  }
  {
    ProfScopeOn ps(ctx, AGZ_PROF_WINO_OUT, st == ctx->stream);
    const size_t n_out = (size_t)a.T * a.Cout_p;
    if (h.raw) {
      const unsigned bd = a.Ntot % 256 == 0 ? 256u : (a.Ntot % 128 == 0 ? 128u : (a.Ntot % 64 == 0 ? 64u : 32u));
      const dim3 gr((unsigned)a.T, (unsigned)a.Ntot / bd);
      if (tm == 5) hipLaunchKernelGGL(wino_out_raw_h2_kernel<5>, gr, dim3(bd), 0, st, h);
      else hipLaunchKernelGGL(wino_out_raw_h2_kernel<4>, gr, dim3(bd), 0, st, h);
    } else if (form == 3 && (a.Cout_p % 256 == 0 || a.Cout_p == 128)) {   // 1 KB runs: 0.251 -> 0.215 ms on the headline block, bit-identical (A/B in profiles/r03/out_quad_ab.log)
      const dim3 gq((unsigned)a.T, (unsigned)ceil_div(a.Cout_p, 256));
      if (tm == 5) hipLaunchKernelGGL(wino_out_quad_h2_kernel<5>, gq, dim3(7 * 64), 0, st, h);
      else hipLaunchKernelGGL(wino_out_quad_h2_kernel<4>, gq, dim3(6 * 64), 0, st, h);
    } else if (form == 3) {
      const unsigned bd = a.Cout_p % 256 == 0 ? 256u : (a.Cout_p % 128 == 0 ? 128u : 64u);
      const dim3 gs((unsigned)a.T, (unsigned)a.Cout_p / bd);
      if (tm == 5) hipLaunchKernelGGL(wino_out_seq_h2_kernel<5>, gs, dim3(bd), 0, st, h);
      else hipLaunchKernelGGL(wino_out_seq_h2_kernel<4>, gs, dim3(bd), 0, st, h);
    } else {
      if (tm == 5) hipLaunchKernelGGL(wino_out_h2_kernel<5>, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, h);
      else hipLaunchKernelGGL(wino_out_h2_kernel<4>, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, st, h);
    }
    if (h.wave_max && h.amax_out && !fuse)   // (fused: the next block's input transform reduces wave_max itself)
      hipLaunchKernelGGL(wino_board_max_kernel, dim3(a.B), dim3(64), 0, st, h.wave_max, h.amax_out, h.wm_per_board);
  }
}
