// Dual-block 3x3 convolution on the fp16 matrix pipe with a 2-way split of both operands ("fp16x2", AGZ_COMPUTE_FP16X2).
//
// x = hi + lo with hi = RN_fp16(x*s), lo = RN_fp16(x*s - hi) (s a power of two) represents x to 2^-23 relative — one
// bit short of fp32 — provided hi and lo stay inside fp16's exponent range; fp16 x fp16 products are exact in the MFMA's
// fp32 accumulation.  Three products per pair (hi*hi, hi*lo, lo*hi; the dropped lo*lo is <= 2^-22) replace the six of
// the bf16x3 formulation (conv_x3.hpp), which is power-bound on its MFMAs: half the matrix instructions, two thirds of
// the operand bytes.  On realistic data the result is as close to the exact dot product as fp32 arithmetic itself
// (K=2304 study in DESIGN.md 4c: rms error 3.0e-7 vs 2.9e-7 for fp32 FMA accumulation, the accumulation dominates).
//
// Range management (what bf16x3 does not need): each tensor is scaled by a power of two so that its largest magnitude
// lands in [2^13, 2^14) — no overflow is possible (fp16 max 65504), hi is a normal number down to 2^-28 of the maximum and
// lo down to 2^-17 of it; smaller elements lose relative (not absolute) precision gracefully.  Weights: scale fixed at
// commit.  Activations: board_amax_kernel reduces max|x| of every BOARD of the layer input (one word per board, so a
// board's result does not depend on what else is in the batch — bitwise), each staging thread scales its row by its
// board's power of two and the epilogue un-scales each output row by the exact inverse.  No host round trip.
//
// Tile 128 x 128 ([64 a | 64 b] columns, shared epilogue), 4 waves x (2x2) MFMA tiles, K step 32 channels of one tap
// (two MFMA k-steps, 24 MFMAs per barrier), chunk-major K order; LDS rows of 64 B, 16-byte chunks XOR-swizzled with
// (row>>2)&3; ONE 32 KB stage with the next tile waiting in registers (two barriers per K step, three workgroups per CU:
// measured 1.36 ms vs 1.39 ms for two stages at two workgroups per CU).   w2[cc32][tap][piece][n][32] fp16.
// Shapes with 2K a multiple of 256 use the wide variant at the end of this file (1.28 ms).
#pragma once
// (included by net.hip INSIDE namespace agz, after conv_x3.hpp)

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned h2_lds_off(int row, int chunk) { return (unsigned)(row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4)); }

// scale exponent from the max-magnitude word: returns s = 2^(13 - floor(log2(amax))) and 1/s (both exact)
__device__ __forceinline__ void h2_scales(unsigned amax_bits, float* s, float* inv) {
  int e = (int)((amax_bits >> 23) & 0xffu);
  if (amax_bits == 0u) { *s = 1.f; *inv = 1.f; return; }
  e = e < 30 ? 30 : (e > 230 ? 230 : e);
  *s = __uint_as_float((unsigned)(267 - e) << 23);
  *inv = __uint_as_float((unsigned)(e - 13) << 23);
}

// max |x| over one board's interior pixels and channels -> amax[b] (float bits; activations are >= 0 but |.| keeps it
// general).  One workgroup per board; the tensor was just written, so this mostly reads the Infinity Cache.
// (t != nullptr: max |x * t[c]| — the Winograd fp16x2 path's equilibrated input range, conv_wino_h2.hpp)
__global__ __launch_bounds__(256) void board_amax_kernel(const float* __restrict__ x, unsigned* __restrict__ amax, int HW, int W, int Wp,
                                                         int HpWp, int C, const float* __restrict__ t = nullptr) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int c4 = C >> 2;
  float m = 0.f;
  for (int i = tid; i < HW * c4; i += 256) {
    int p = i / c4, c = (i - p * c4) << 2;
    int h = p / W, w = p - h * W;
    float4 v = *reinterpret_cast<const float4*>(x + ((size_t)b * HpWp + (h + 1) * Wp + (w + 1)) * C + c);
    if (t) { const float4 tt = *reinterpret_cast<const float4*>(t + c); v.x *= tt.x; v.y *= tt.y; v.z *= tt.z; v.w *= tt.w; }
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) amax[b] = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
}

// the same reduction spread over `parts` workgroups per board (grid = B * parts, amax zeroed by the caller): the trainer's
// 256-board passes, where one workgroup per board leaves the chip reading at 0.9 TB/s.  max is order-independent: same words.
__global__ __launch_bounds__(256) void board_amax_parts_kernel(const float* __restrict__ x, unsigned* __restrict__ amax, int HW, int W, int Wp,
                                                               int HpWp, int C, int parts) {
  __shared__ float red[4];
  const int b = blockIdx.x / parts, part = blockIdx.x - b * parts, tid = threadIdx.x;
  const int c4 = C >> 2;
  const int n = HW * c4, lo = (int)((long)n * part / parts), hi = (int)((long)n * (part + 1) / parts);
  float m = 0.f;
  for (int i = lo + tid; i < hi; i += 256) {
    int p = i / c4, c = (i - p * c4) << 2;
    int h = p / W, w = p - h * W;
    const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)b * HpWp + (h + 1) * Wp + (w + 1)) * C + c);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) atomicMax(&amax[b], __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

__global__ __launch_bounds__(256, 3) void conv3x3_h2_kernel(ConvArgs a, const _Float16* __restrict__ w2) {
  constexpr int BM = 128, BNT = 128;
  constexpr int PIECE = 128 * 64;            // bytes of one piece image (128 rows x 32 fp16)
  constexpr int STAGE = 4 * PIECE;           // A hi, A lo, B hi, B lo
  __shared__ __attribute__((aligned(16))) unsigned char lds[STAGE];

  const int nblk = a.n_mtiles * a.n_ntiles;
  const int id = blockIdx.x;
  int q = nblk >> 3, rr = nblk & 7, xcd = id & 7, slot = id >> 3;
  int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + slot;
  const int m_tile = tile / a.n_ntiles, n_tile = tile - m_tile * a.n_ntiles;
  const int m0 = m_tile * BM, n0 = n_tile * BNT;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;


  // ---- staging: thread t owns (row t/2, 16-channel half t%2) of both operand tiles
  const int srow = tid >> 1, shalf = tid & 1;
  unsigned a_gbyte;
  float sa;
  {
    int m = m0 + srow;
    if (m >= a.M) m = a.M - 1;
    int b = m / a.HW, p = m - b * a.HW;
    int h = p / a.W, w = p - h * a.W;
    a_gbyte = (unsigned)((((b * a.HpWp) + (h + 1) * a.Wp + (w + 1)) * a.Cin_p + shalf * 16) * 4);
    float inv_;
    h2_scales(a.amax_in[b], &sa, &inv_);      // the scale of this thread's staging row = of its board
  }
  int nrow = n0 + srow;
  if (nrow >= a.Ntot) nrow = a.Ntot - 1;
  const unsigned b_gbyte = (unsigned)(nrow * 64 + shalf * 32);
  const unsigned piece_bytes = (unsigned)a.Ntot * 64u;
  const unsigned s_off0 = h2_lds_off(srow, 2 * shalf), s_off1 = h2_lds_off(srow, 2 * shalf + 1);
  const int NC = a.Cin_p >> 5;               // 32-channel chunks
  const int NK = 9 * NC;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // fragment rows; chunk = 2*kstep + (lane>>5)
  int ra[2], rb[2];
#pragma unroll
  for (int i = 0; i < 2; i++) ra[i] = (wm * 2 + i) * 32 + (lane & 31);
  rb[0] = wn * 32 + (lane & 31);          // branch a channels
  rb[1] = 64 + wn * 32 + (lane & 31);     // matching branch b channels
  const int kh = lane >> 5;

  float4 xa0, xa1, xa2, xa3;
  u32x4_t xb0, xb1, xb2, xb3;   // B: hi (2 chunks), lo (2 chunks)
  const char* xbase = reinterpret_cast<const char*>(a.x);
  const char* wbase = reinterpret_cast<const char*>(w2);
  int f_kx = 0, f_ky = 0, f_n = 0;
  const unsigned row_step = (unsigned)(a.Cin_p * 4), line_step = (unsigned)((a.Wp - 2) * a.Cin_p * 4);
  const unsigned chunk_back = (unsigned)((2 * a.Wp + 2) * a.Cin_p * 4) - 128u;
  unsigned xo_ = a_gbyte - (unsigned)((a.Wp + 1) * a.Cin_p * 4), wo_ = b_gbyte;
#define H2_GLOAD()                                                                            \
  xa0 = *reinterpret_cast<const float4*>(xbase + xo_);                                        \
  xa1 = *reinterpret_cast<const float4*>(xbase + xo_ + 16u);                                  \
  xa2 = *reinterpret_cast<const float4*>(xbase + xo_ + 32u);                                  \
  xa3 = *reinterpret_cast<const float4*>(xbase + xo_ + 48u);                                  \
  xb0 = *reinterpret_cast<const u32x4_t*>(wbase + wo_);                                       \
  xb1 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + 16u);                                 \
  xb2 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + piece_bytes);                         \
  xb3 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + piece_bytes + 16u);                   \
  if (f_n + 1 < NK) {                                                                         \
    f_n++;                                                                                    \
    wo_ += 2u * piece_bytes;                                                                  \
    if (f_kx < 2) { f_kx++; xo_ += row_step; }                                                \
    else if (f_ky < 2) { f_kx = 0; f_ky++; xo_ += line_step; }                                \
    else { f_kx = 0; f_ky = 0; xo_ -= chunk_back; }                                           \
  }
#define H2_SPLIT8(V0, V1, HI, LO)                                                             \
  {                                                                                           \
    const float v_[8] = {V0.x, V0.y, V0.z, V0.w, V1.x, V1.y, V1.z, V1.w};                     \
    _Pragma("unroll") for (int e = 0; e < 8; e++) {                                           \
      float xs_ = v_[e] * sa;                                                                 \
      _Float16 h_ = (_Float16)xs_;                                                            \
      HI[e] = h_;                                                                             \
      LO[e] = (_Float16)(xs_ - (float)h_);                                                    \
    }                                                                                         \
  }
#define H2_STORE(BUF)                                                                         \
  {                                                                                           \
    unsigned char* st_ = lds;                                                 \
    f16x8_t h0_, l0_, h1_, l1_;                                                               \
    H2_SPLIT8(xa0, xa1, h0_, l0_)                                                             \
    H2_SPLIT8(xa2, xa3, h1_, l1_)                                                             \
    *reinterpret_cast<f16x8_t*>(st_ + 0 * PIECE + s_off0) = h0_;                              \
    *reinterpret_cast<f16x8_t*>(st_ + 0 * PIECE + s_off1) = h1_;                              \
    *reinterpret_cast<f16x8_t*>(st_ + 1 * PIECE + s_off0) = l0_;                              \
    *reinterpret_cast<f16x8_t*>(st_ + 1 * PIECE + s_off1) = l1_;                              \
    *reinterpret_cast<u32x4_t*>(st_ + 2 * PIECE + s_off0) = xb0;                              \
    *reinterpret_cast<u32x4_t*>(st_ + 2 * PIECE + s_off1) = xb1;                              \
    *reinterpret_cast<u32x4_t*>(st_ + 3 * PIECE + s_off0) = xb2;                              \
    *reinterpret_cast<u32x4_t*>(st_ + 3 * PIECE + s_off1) = xb3;                              \
  }
  // one piece product on the four sub-tiles
#define H2_QUAD(PA_, PB_)                                                                     \
  _Pragma("unroll") for (int i = 0; i < 2; i++)                                               \
    _Pragma("unroll") for (int j = 0; j < 2; j++)                                             \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[i][PA_], B_[j][PB_], acc[i][j], 0, 0, 0);

#define H2_COMPUTE(ST)                                                                        \
  _Pragma("unroll") for (int ks = 0; ks < 2; ks++) {                                          \
    f16x8_t A_[2][2], B_[2][2];                                                               \
    _Pragma("unroll") for (int p = 0; p < 2; p++) {                                           \
      _Pragma("unroll") for (int i = 0; i < 2; i++) A_[i][p] = *reinterpret_cast<const f16x8_t*>((ST) + p * PIECE + h2_lds_off(ra[i], 2 * ks + kh));       \
      _Pragma("unroll") for (int j = 0; j < 2; j++) B_[j][p] = *reinterpret_cast<const f16x8_t*>((ST) + (2 + p) * PIECE + h2_lds_off(rb[j], 2 * ks + kh)); \
    }                                                                                         \
    H2_QUAD(1, 0) H2_QUAD(0, 1) H2_QUAD(0, 0) /* small terms first */                         \
  }
  {
    // one 32 KB stage, the next tile waits in registers: two barriers per K step, but three workgroups per CU
    H2_GLOAD()
    for (int it = 0; it < NK; it++) {
      H2_STORE(0)
      __syncthreads();
      H2_GLOAD()
      H2_COMPUTE(lds)
      __syncthreads();
    }
  }
#undef H2_COMPUTE
#undef H2_QUAD
#undef H2_STORE
#undef H2_SPLIT8
#undef H2_GLOAD

  // --- epilogue: un-scale (exact), BN(scale,shift) + ReLU on both branches, add, ReLU; track max(output) for the next layer
#pragma unroll
  for (int i = 0; i < 2; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      int row = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      int m = m0 + row;
      const bool mvalid = m < a.M;
      if (!mvalid) m = a.M - 1;
      int b = m / a.HW, p = m - b * a.HW;
      int h = p / a.W, w = p - h * a.W;
      size_t obase = ((size_t)b * a.HpWp + (h + 1) * a.Wp + (w + 1)) * a.Cout_p;
      float s_, unscale;
      h2_scales(a.amax_in[b], &s_, &unscale);
      unscale *= a.w_unscale;
      int c = n_tile * (BNT / 2) + wn * 32 + (lane & 31);
      if (mvalid && c < a.Cout_p) {
        float4 e = reinterpret_cast<const float4*>(a.ep)[(size_t)p * a.Cout_p + c];
        float va = (acc[i][0][r] * unscale) * e.x + e.y;
        float vb = (acc[i][1][r] * unscale) * e.z + e.w;
        va = va > 0.f ? va : 0.f;
        vb = vb > 0.f ? vb : 0.f;
        float s = va + vb;
        a.y[obase + c] = s > 0.f ? s : 0.f;
      }
    }
  }
}

// ---- wide variant: 128 x 256 block tile, 64 x 128 per wave (8 accumulators), one 48 KB stage, 2 workgroups/CU ----------
// With half the MFMAs of bf16x3 this formulation is issue-bound rather than power-bound (1.36 ms, 1.17 on zeros, MFMA-only
// 0.52): the fix is fewer non-MFMA instructions per MFMA.  Per wave and K step (32 channels): 48 MFMAs for 24 fragment
// reads (0.5/MFMA instead of 0.67), 12 ds_write_b128 (0.25 instead of 0.33), and the activation split (the bulk of the
// VALU work) is amortised over twice the columns.  256 GEMM columns = two [64 a | 64 b] groups of the same w2 layout;
// wave column wn owns group wn.
__global__ __launch_bounds__(256, 2) void conv3x3_h2w_kernel(ConvArgs a, const _Float16* __restrict__ w2) {
  constexpr int BM = 128, BNT = 256;
  constexpr int PA = 128 * 64, PB = 256 * 64;     // bytes of one A / B piece image
  constexpr int STAGE = 2 * PA + 2 * PB;          // 48 KB
  __shared__ __attribute__((aligned(16))) unsigned char lds[STAGE];

  const int nblk = a.n_mtiles * a.n_ntiles;
  const int id = blockIdx.x;
  int q = nblk >> 3, rr = nblk & 7, xcd = id & 7, slot = id >> 3;
  int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + slot;
  const int m_tile = tile / a.n_ntiles, n_tile = tile - m_tile * a.n_ntiles;
  const int m0 = m_tile * BM, n0 = n_tile * BNT;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;


  const int srow = tid >> 1, shalf = tid & 1;
  unsigned a_gbyte;
  float sa;
  {
    int m = m0 + srow;
    if (m >= a.M) m = a.M - 1;
    int b = m / a.HW, p = m - b * a.HW;
    int h = p / a.W, w = p - h * a.W;
    a_gbyte = (unsigned)((((b * a.HpWp) + (h + 1) * a.Wp + (w + 1)) * a.Cin_p + shalf * 16) * 4);
    float inv_;
    h2_scales(a.amax_in[b], &sa, &inv_);      // the scale of this thread's staging row = of its board
  }
  int nr0 = n0 + srow, nr1 = n0 + 128 + srow;
  if (nr0 >= a.Ntot) nr0 = a.Ntot - 1;
  if (nr1 >= a.Ntot) nr1 = a.Ntot - 1;
  const unsigned b_g0 = (unsigned)(nr0 * 64 + shalf * 32), b_g1 = (unsigned)(nr1 * 64 + shalf * 32);
  const unsigned piece_bytes = (unsigned)a.Ntot * 64u;
  const unsigned sa0 = h2_lds_off(srow, 2 * shalf), sa1 = h2_lds_off(srow, 2 * shalf + 1);
  const unsigned sb0 = h2_lds_off(128 + srow, 2 * shalf), sb1 = h2_lds_off(128 + srow, 2 * shalf + 1);
  const int NC = a.Cin_p >> 5;
  const int NK = 9 * NC;

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
  // columns of this wave inside the 256-column tile: group wn = [64 a | 64 b]; j = 0,1 branch a, j = 2,3 branch b
#pragma unroll
  for (int j = 0; j < 4; j++) rb[j] = wn * 128 + (j >> 1) * 64 + (j & 1) * 32 + (lane & 31);
  const int kh = lane >> 5;

  float4 xa0, xa1, xa2, xa3;
  u32x4_t xb0, xb1, xb2, xb3, xb4, xb5, xb6, xb7;   // B: rows srow (hi 2, lo 2), rows 128+srow (hi 2, lo 2)
  const char* xbase = reinterpret_cast<const char*>(a.x);
  const char* wbase = reinterpret_cast<const char*>(w2);
  int f_kx = 0, f_ky = 0, f_n = 0;
  const unsigned row_step = (unsigned)(a.Cin_p * 4), line_step = (unsigned)((a.Wp - 2) * a.Cin_p * 4);
  const unsigned chunk_back = (unsigned)((2 * a.Wp + 2) * a.Cin_p * 4) - 128u;
  unsigned xo_ = a_gbyte - (unsigned)((a.Wp + 1) * a.Cin_p * 4), wo_ = 0;
#define H2W_GLOAD()                                                                           \
  xa0 = *reinterpret_cast<const float4*>(xbase + xo_);                                        \
  xa1 = *reinterpret_cast<const float4*>(xbase + xo_ + 16u);                                  \
  xa2 = *reinterpret_cast<const float4*>(xbase + xo_ + 32u);                                  \
  xa3 = *reinterpret_cast<const float4*>(xbase + xo_ + 48u);                                  \
  xb0 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + b_g0);                                \
  xb1 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + b_g0 + 16u);                          \
  xb2 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + piece_bytes + b_g0);                  \
  xb3 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + piece_bytes + b_g0 + 16u);            \
  xb4 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + b_g1);                                \
  xb5 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + b_g1 + 16u);                          \
  xb6 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + piece_bytes + b_g1);                  \
  xb7 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + piece_bytes + b_g1 + 16u);            \
  if (f_n + 1 < NK) {                                                                         \
    f_n++;                                                                                    \
    wo_ += 2u * piece_bytes;                                                                  \
    if (f_kx < 2) { f_kx++; xo_ += row_step; }                                                \
    else if (f_ky < 2) { f_kx = 0; f_ky++; xo_ += line_step; }                                \
    else { f_kx = 0; f_ky = 0; xo_ -= chunk_back; }                                           \
  }
#define H2W_SPLIT8(V0, V1, HI, LO)                                                            \
  {                                                                                           \
    const float v_[8] = {V0.x, V0.y, V0.z, V0.w, V1.x, V1.y, V1.z, V1.w};                     \
    _Pragma("unroll") for (int e = 0; e < 8; e++) {                                           \
      float xs_ = v_[e] * sa;                                                                 \
      _Float16 h_ = (_Float16)xs_;                                                            \
      HI[e] = h_;                                                                             \
      LO[e] = (_Float16)(xs_ - (float)h_);                                                    \
    }                                                                                         \
  }
#define H2W_STORE()                                                                           \
  {                                                                                           \
    f16x8_t h0_, l0_, h1_, l1_;                                                               \
    H2W_SPLIT8(xa0, xa1, h0_, l0_)                                                            \
    H2W_SPLIT8(xa2, xa3, h1_, l1_)                                                            \
    *reinterpret_cast<f16x8_t*>(lds + 0 * PA + sa0) = h0_;                                    \
    *reinterpret_cast<f16x8_t*>(lds + 0 * PA + sa1) = h1_;                                    \
    *reinterpret_cast<f16x8_t*>(lds + 1 * PA + sa0) = l0_;                                    \
    *reinterpret_cast<f16x8_t*>(lds + 1 * PA + sa1) = l1_;                                    \
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
#define H2W_OCT(PA_, PB_)                                                                     \
  _Pragma("unroll") for (int i = 0; i < 2; i++)                                               \
    _Pragma("unroll") for (int j = 0; j < 4; j++)                                             \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[i][PA_], B_[j][PB_], acc[i][j], 0, 0, 0);

  H2W_GLOAD()
  for (int it = 0; it < NK; it++) {
    H2W_STORE()
    __syncthreads();
    H2W_GLOAD()
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
      H2W_OCT(1, 0) H2W_OCT(0, 1) H2W_OCT(0, 0)
    }
    __syncthreads();
  }
#undef H2W_OCT
#undef H2W_STORE
#undef H2W_SPLIT8
#undef H2W_GLOAD

#pragma unroll
  for (int i = 0; i < 2; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      int row = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      int m = m0 + row;
      const bool mvalid = m < a.M;
      if (!mvalid) m = a.M - 1;
      int b = m / a.HW, p = m - b * a.HW;
      int h = p / a.W, w = p - h * a.W;
      size_t obase = ((size_t)b * a.HpWp + (h + 1) * a.Wp + (w + 1)) * a.Cout_p;
      float s_, unscale;
      h2_scales(a.amax_in[b], &s_, &unscale);
      unscale *= a.w_unscale;
      if (a.raw) {
        // training forward (conv3x3_raw_h2): the GEMM result as is, columns in the weight image's own order; the weights' scale is a
        // device word (they change every step)
        float sw_, unw_;
        h2_scales(*a.w_amax_dev, &sw_, &unw_);
        unscale *= unw_;
        float* yr = a.y + ((size_t)b * a.HpWp + (h + 1) * a.Wp + (w + 1)) * a.Ntot;
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int c = n0 + rb[j];
          if (mvalid && c < a.Ntot) yr[c] = acc[i][j][r] * unscale;
        }
        continue;
      }
#pragma unroll
      for (int jj = 0; jj < 2; jj++) {
        int c = n_tile * 128 + wn * 64 + jj * 32 + (lane & 31);
        if (mvalid && c < a.Cout_p) {
          float4 e = reinterpret_cast<const float4*>(a.ep)[(size_t)p * a.Cout_p + c];
          float va = (acc[i][jj][r] * unscale) * e.x + e.y;
          float vb = (acc[i][2 + jj][r] * unscale) * e.z + e.w;
          va = va > 0.f ? va : 0.f;
          vb = vb > 0.f ? vb : 0.f;
          float s = va + vb;
          a.y[obase + c] = s > 0.f ? s : 0.f;
        }
      }
    }
  }
}

