// Dual-block 3x3 convolution with fp32-grade arithmetic on the bf16 matrix pipe ("bf16x3").
//
// v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 MFMA rate (157 vs 2517 TFLOP/s).  Every fp32 value splits EXACTLY
// into three bf16 pieces by truncation (24-bit significand = 8 + 8 + 8: hi = top 8 bits, mid = top 8 bits of the exact
// remainder, lo = the exact rest), and products of bf16 pairs are exact in the MFMA's fp32 accumulation.  Of the nine
// piece products the kernel computes the six of relative weight >= 2^-16:
//     a*b ~= hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi        (dropped: mid*lo, lo*mid, lo*lo <= 2^-23 |a||b|)
// i.e. the dropped part is below the rounding error of a single fp32 multiply.  Six bf16 MFMAs replace sixteen
// fp32-MFMA passes: the roofline of this formulation is 2516.6 / 6 = 419 algorithmic TFLOP/s (2.67x the fp32 pipe).
//
// Weights are pre-split at commit time (w3 layout below); activations stay fp32 in HBM and are split in registers on
// their way into LDS (4 VALU ops per element, hidden behind the MFMAs).
//
// Tile: 128 pixels x 128 GEMM columns ([64 branch-a | 64 branch-b] channels, same interleave as the fp32 kernel, so the
// fused BN/ReLU/add epilogue is shared), 4 waves (2x2) of 64x64, K step 16 channels of one filter tap per iteration.
// LDS per stage: A 3 pieces x 128 rows x 32 B + B the same = 24 KB; two stages.
//   w3[cc][tap][piece][n][16] bf16: the 128 x 16 B-tile of one (cc, tap, piece) is one contiguous 4 KB run, and the
//   K loop (chunk outer, tap inner) walks the array front to back.
//
// Measured (MI355X, 19x19, K=256, 512 boards): 2.0-2.1 ms per dual block vs 3.23 ms for the fp32-MFMA kernel.  The
// kernel is POWER-bound, not issue-bound: with all-zero weights the identical instruction stream runs in 1.70 ms, a
// bare bf16 MFMA loop reaches 2130 TFLOP/s on non-zero data (2464 on zeros), and making the global loads cache-hot,
// dropping the split, dropping the barrier, bypassing LDS for the weights, pinning the schedule, or a 128x256 block
// tile with 64x128 per wave (25 % fewer LDS reads per MFMA) all land within +-4 % of each other: the MFMAs themselves
// on real (toggling) data draw most of the power budget — a bare MFMA loop over random rotating operands sustains
// 1772 TFLOP/s bf16 = 295 in algorithmic units, and this kernel runs at 0.70 of that.
#pragma once
// (included by net.hip INSIDE namespace agz, after ConvArgs / f32x16)

typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void x3_split(float v, unsigned& h, unsigned& m, unsigned& l) {
  unsigned hu = __float_as_uint(v) & 0xffff0000u;
  float r = v - __uint_as_float(hu);
  unsigned mu = __float_as_uint(r) & 0xffff0000u;
  float r2 = r - __uint_as_float(mu);
  h = hu; m = mu; l = __float_as_uint(r2);
}
// two fp32 bit patterns whose low halves are dead -> packed bf16 pair (e0 in the low half)
__device__ __forceinline__ unsigned x3_pack(unsigned e0, unsigned e1) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }

// 16-byte chunk position of (row, half) inside a piece image of 32-byte rows: XOR with bit 3 of the row makes both the
// ds_write_b128 (rows t/2, halves t%2) and the fragment ds_read_b128 (32 rows x fixed half) conflict-free
__device__ __forceinline__ unsigned x3_lds_off(int row, int half) { return (unsigned)(row * 32 + ((half ^ ((row >> 3) & 1)) << 4)); }

// the 32 lanes of a half wave hold one row's columns: their maximum goes to the row's board with one atomic (outputs are >= 0 after
// the ReLU, so the float's bits order like the value: the word board_amax_kernel computes)
__device__ __forceinline__ void x3_amax_flush(unsigned* amax, int b, float m, int lane) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((lane & 31) == 0) atomicMax(&amax[b], __float_as_uint(m));
}
// shared epilogue (same as conv_tile): BN(scale,shift) + ReLU (+ dual add + ReLU), interior of padded NHWC
#define X3_EPILOGUE \
  int amx_b = -1; float amx_m = 0.f;   /* running maximum of this thread's outputs of board amx_b (rows ascend: so do boards) */ \
  _Pragma("unroll") \
  for (int i = 0; i < 2; i++) { \
  _Pragma("unroll") \
    for (int r = 0; r < 16; r++) { \
      int row = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); \
      int m = m0 + row; \
      const bool mvalid = m < a.M; \
      if (!mvalid) m = a.M - 1; \
      int b = m / a.HW, p = m - b * a.HW; \
      int h = p / a.W, w = p - h * a.W; \
      size_t obase = ((size_t)b * a.HpWp + (h + 1) * a.Wp + (w + 1)) * a.Cout_p; \
      if (DUAL) { \
        int c = n_tile * (BNT / 2) + wn * 32 + (lane & 31); \
        if (mvalid && c < a.Cout_p) { \
          float4 e = reinterpret_cast<const float4*>(a.ep)[(size_t)p * a.Cout_p + c]; \
          float va = acc[i][0][r] * e.x + e.y; \
          float vb = acc[i][1][r] * e.z + e.w; \
          va = va > 0.f ? va : 0.f; \
          vb = vb > 0.f ? vb : 0.f; \
          float s = va + vb; \
          a.y[obase + c] = s > 0.f ? s : 0.f; \
        } \
      } else { \
        if (a.amax_out && mvalid && b != amx_b) {   /* (uniform over the 32 lanes that share this row) */ \
          if (amx_b >= 0) x3_amax_flush(a.amax_out, amx_b, amx_m, lane); \
          amx_b = b; amx_m = 0.f; \
        } \
  _Pragma("unroll") \
        for (int j = 0; j < 2; j++) { \
          int c = n0 + (wn * 2 + j) * 32 + (lane & 31); \
          if (mvalid && c < a.Cout_p) { \
            if (a.raw) { \
              a.y[obase + c] = acc[i][j][r]; \
            } else { \
              float2 e = reinterpret_cast<const float2*>(a.ep)[(size_t)p * a.Cout_p + c]; \
              float v = acc[i][j][r] * e.x + e.y; \
              v = v > 0.f ? v : 0.f; \
              a.y[obase + c] = v; \
              amx_m = fmaxf(amx_m, v); \
            } \
          } \
        } \
      } \
    } \
  } \
  if (!DUAL && a.amax_out && amx_b >= 0) x3_amax_flush(a.amax_out, amx_b, amx_m, lane);

template <bool DUAL>
__global__ __launch_bounds__(256, 3) void conv3x3_x3_kernel(ConvArgs a, const unsigned short* __restrict__ w3) {
  constexpr int BM = 128, BNT = 128;
  constexpr int PIECE = 128 * 32;            // bytes of one piece image
  constexpr int STAGE = 6 * PIECE;           // A hi,mid,lo then B hi,mid,lo
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

  const int nblk = a.n_mtiles * a.n_ntiles;
  const int id = blockIdx.x;
  int q = nblk >> 3, rr = nblk & 7, xcd = id & 7, slot = id >> 3;
  int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + slot;
  const int m_tile = tile / a.n_ntiles, n_tile = tile - m_tile * a.n_ntiles;
  const int m0 = m_tile * BM, n0 = n_tile * BNT;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;

  // ---- staging: thread t owns (row t/2, 8-channel half t%2) of both operand tiles
  const int srow = tid >> 1, shalf = tid & 1;
  int a_goff;
  {
    int m = m0 + srow;
    if (m >= a.M) m = a.M - 1;
    int b = m / a.HW, p = m - b * a.HW;
    int h = p / a.W, w = p - h * a.W;
    a_goff = ((b * a.HpWp) + (h + 1) * a.Wp + (w + 1)) * a.Cin_p + shalf * 8;
  }
  int nrow = n0 + srow;
  if (nrow >= a.Ntot) nrow = a.Ntot - 1;
  const int b_goff = nrow * 16 + shalf * 8;                 // bf16 elements inside one [n][16] piece tile
  const size_t piece_stride = (size_t)a.Ntot * 16;          // bf16 elements
  const unsigned s_off = x3_lds_off(srow, shalf);
  const int NC = a.Cin_p >> 4;                               // 16-channel chunks per tap
  const int NK = 9 * NC;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // fragment addresses (bytes inside a piece image)
  unsigned fa[2], fb[2];
#pragma unroll
  for (int i = 0; i < 2; i++) fa[i] = x3_lds_off((wm * 2 + i) * 32 + (lane & 31), lane >> 5);
  if (DUAL) {
    fb[0] = x3_lds_off(wn * 32 + (lane & 31), lane >> 5);        // branch a channels
    fb[1] = x3_lds_off(64 + wn * 32 + (lane & 31), lane >> 5);   // the matching branch b channels
  } else {
    fb[0] = x3_lds_off((wn * 2 + 0) * 32 + (lane & 31), lane >> 5);
    fb[1] = x3_lds_off((wn * 2 + 1) * 32 + (lane & 31), lane >> 5);
  }

  // staging registers: two named sets (tile it+1 waits in one while tile it+2 is being fetched into the other)
  float4 xa0, xa1, ya0, ya1;           // A: 8 fp32
  u32x4_t xb0, xb1, xb2, yb0, yb1, yb2;  // B: 3 pieces x 8 bf16
  // fetch cursor: (tap, 16-channel chunk) of the NEXT tile to load, advanced incrementally (no per-iteration division);
  // 32-bit byte offsets from the two uniform base pointers keep the address arithmetic off the VALU
  const char* xbase = reinterpret_cast<const char*>(a.x);
  const char* wbase = reinterpret_cast<const char*>(w3);
  const unsigned a_gbyte = (unsigned)a_goff * 4u, b_gbyte = (unsigned)b_goff * 2u;
  const unsigned piece_bytes = (unsigned)(piece_stride * 2);
  // K order: 16-channel chunk OUTER, filter tap INNER — the nine taps of a chunk re-read the same 64-byte slices of
  // ~170 neighbouring pixels back to back, so they hit in L1/L2 instead of going back to the Infinity Cache nine times
  // (tap-major order measured 3.2 GB of L2 fills per launch for 0.2 GB of activations).
  int f_kx = 0, f_ky = 0, f_n = 0;
  const unsigned row_step = (unsigned)(a.Cin_p * 4), line_step = (unsigned)((a.Wp - 2) * a.Cin_p * 4);
  const unsigned chunk_back = (unsigned)((2 * a.Wp + 2) * a.Cin_p * 4) - 64u;   // from tap (2,2) back to tap (0,0), next chunk
  unsigned xo_ = a_gbyte - (unsigned)((a.Wp + 1) * a.Cin_p * 4), wo_ = b_gbyte;
#define X3_ADVANCE()                                                                          \
  if (f_n + 1 < NK) {  /* past the end: keep re-reading the last tile (never consumed) */     \
    f_n++;                                                                                    \
    wo_ += 3u * piece_bytes;                                                                  \
    if (f_kx < 2) { f_kx++; xo_ += row_step; }                                                \
    else if (f_ky < 2) { f_kx = 0; f_ky++; xo_ += line_step; }                                \
    else { f_kx = 0; f_ky = 0; xo_ -= chunk_back; }                                           \
  }
#define X3_GLOAD(A0, A1, B0, B1, B2)                                                          \
  A0 = *reinterpret_cast<const float4*>(xbase + xo_);                                         \
  A1 = *reinterpret_cast<const float4*>(xbase + xo_ + 16u);                                   \
  B0 = *reinterpret_cast<const u32x4_t*>(wbase + wo_);                                        \
  B1 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + piece_bytes);                          \
  B2 = *reinterpret_cast<const u32x4_t*>(wbase + wo_ + 2u * piece_bytes);                     \
  X3_ADVANCE()
#define X3_STORE_A(A0, A1, BUF)                                                              \
  {                                                                                          \
    unsigned char* st_ = lds + (BUF) * STAGE;                                                \
    const float v_[8] = {A0.x, A0.y, A0.z, A0.w, A1.x, A1.y, A1.z, A1.w};                    \
    unsigned h_[8], m_[8], l_[8];                                                            \
    _Pragma("unroll") for (int e = 0; e < 8; e++) x3_split(v_[e], h_[e], m_[e], l_[e]);      \
    u32x4_t ph_ = {x3_pack(h_[0], h_[1]), x3_pack(h_[2], h_[3]), x3_pack(h_[4], h_[5]), x3_pack(h_[6], h_[7])}; \
    u32x4_t pm_ = {x3_pack(m_[0], m_[1]), x3_pack(m_[2], m_[3]), x3_pack(m_[4], m_[5]), x3_pack(m_[6], m_[7])}; \
    u32x4_t pl_ = {x3_pack(l_[0], l_[1]), x3_pack(l_[2], l_[3]), x3_pack(l_[4], l_[5]), x3_pack(l_[6], l_[7])}; \
    *reinterpret_cast<u32x4_t*>(st_ + 0 * PIECE + s_off) = ph_;                              \
    *reinterpret_cast<u32x4_t*>(st_ + 1 * PIECE + s_off) = pm_;                              \
    *reinterpret_cast<u32x4_t*>(st_ + 2 * PIECE + s_off) = pl_;                              \
  }
#define X3_STORE_B(B0, B1, B2, BUF)                                                          \
  {                                                                                          \
    unsigned char* st_ = lds + (BUF) * STAGE;                                                \
    *reinterpret_cast<u32x4_t*>(st_ + 3 * PIECE + s_off) = B0;                               \
    *reinterpret_cast<u32x4_t*>(st_ + 4 * PIECE + s_off) = B1;                               \
    *reinterpret_cast<u32x4_t*>(st_ + 5 * PIECE + s_off) = B2;                               \
  }
#define X3_SB   /* no sched_barrier pins: measured 1.99 ms with hipcc's own schedule vs 2.05-2.10 pinned */
#define X3_MF(I, J, PA, PB) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[I][PA], B_[J][PB], acc[I][J], 0, 0, 0);
  // one piece product on all four 32x32 sub-tiles: consecutive MFMAs hit different accumulators
#define X3_QUAD(PA, PB) X3_MF(0, 0, PA, PB) X3_MF(0, 1, PA, PB) X3_MF(1, 0, PA, PB) X3_MF(1, 1, PA, PB)
  // One K step.  X* = tile it+1 (already in registers) -> split + ds_write into the other stage;  Y* = tile it+2, fetched now.
  // Memory instructions are slotted between MFMA groups and pinned there: a wave issues in order, so whatever sits
  // behind an MFMA is free while the matrix pipe works; left to itself hipcc sinks the global loads to the end of the
  // iteration and exposes their whole latency at the top of the next one.  Products: smallest terms first.
#define X3_ITER(BUF, XA0, XA1, XB0, XB1, XB2, YA0, YA1, YB0, YB1, YB2)                       \
  {                                                                                          \
    const unsigned char* st = lds + (BUF) * STAGE;                                           \
    bf16x8_t A_[2][3], B_[2][3];                                                             \
    _Pragma("unroll") for (int p = 0; p < 3; p++) {                                          \
      _Pragma("unroll") for (int i = 0; i < 2; i++) A_[i][p] = *reinterpret_cast<const bf16x8_t*>(st + p * PIECE + fa[i]);       \
      _Pragma("unroll") for (int j = 0; j < 2; j++) B_[j][p] = *reinterpret_cast<const bf16x8_t*>(st + (3 + p) * PIECE + fb[j]); \
    }                                                                                        \
    X3_SB                                                                                    \
    X3_QUAD(2, 0) X3_SB                                                                      \
    X3_GLOAD(YA0, YA1, YB0, YB1, YB2) X3_SB                                                  \
    X3_QUAD(0, 2) X3_SB                                                                      \
    X3_STORE_A(XA0, XA1, (BUF) ^ 1) X3_SB                                                    \
    X3_QUAD(1, 1) X3_QUAD(1, 0) X3_SB                                                        \
    X3_STORE_B(XB0, XB1, XB2, (BUF) ^ 1) X3_SB                                               \
    X3_QUAD(0, 1) X3_QUAD(0, 0) X3_SB                                                        \
    __syncthreads();                                                                         \
  }

  // prologue: tile 0 -> stage 0 (through set Y), tile 1 -> set X
  X3_GLOAD(ya0, ya1, yb0, yb1, yb2)
  X3_GLOAD(xa0, xa1, xb0, xb1, xb2)
  X3_STORE_A(ya0, ya1, 0)
  X3_STORE_B(yb0, yb1, yb2, 0)
  __syncthreads();
  for (int it = 0; it < NK; it += 2) {
    X3_ITER(0, xa0, xa1, xb0, xb1, xb2, ya0, ya1, yb0, yb1, yb2)
    if (it + 1 < NK) X3_ITER(1, ya0, ya1, yb0, yb1, yb2, xa0, xa1, xb0, xb1, xb2)
  }
  // (the X3_* staging / pipeline macros stay defined for conv_wino.hpp, included next, which #undefs them)

  X3_EPILOGUE
}


