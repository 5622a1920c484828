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
//   w3[tap][cc][piece][n][16] bf16: the 128 x 16 B-tile of one (tap, cc, piece) is one contiguous 4 KB run.
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
__device__ __forceinline__ unsigned x3_pack(unsigned e0, unsigned e1) { return (e0 >> 16) | (e1 & 0xffff0000u); }

// 16-byte chunk position of (row, half) inside a piece image of 32-byte rows: XOR with bit 3 of the row makes both the
// ds_write_b128 (rows t/2, halves t%2) and the fragment ds_read_b128 (32 rows x fixed half) conflict-free
__device__ __forceinline__ unsigned x3_lds_off(int row, int half) { return (unsigned)(row * 32 + ((half ^ ((row >> 3) & 1)) << 4)); }

template <bool DUAL>
__global__ __launch_bounds__(256, 2) void conv3x3_x3_kernel(ConvArgs a, const unsigned short* __restrict__ w3) {
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

  float4 ga0, ga1;       // A: 8 fp32
  u32x4_t gb0, gb1, gb2;  // B: 3 pieces x 8 bf16

  auto gload = [&](int it) {
    int t = it < NK ? it : NK - 1;
    int tap = t / NC, cc = t - tap * NC;
    int ky = tap / 3, kx = tap - ky * 3;
    const float* xp = a.x + a_goff + ((ky - 1) * a.Wp + (kx - 1)) * a.Cin_p + cc * 16;
    ga0 = *reinterpret_cast<const float4*>(xp);
    ga1 = *reinterpret_cast<const float4*>(xp + 4);
    const unsigned short* wp = w3 + (size_t)(tap * NC + cc) * 3 * piece_stride + b_goff;
    gb0 = *reinterpret_cast<const u32x4_t*>(wp);
    gb1 = *reinterpret_cast<const u32x4_t*>(wp + piece_stride);
    gb2 = *reinterpret_cast<const u32x4_t*>(wp + 2 * piece_stride);
  };
  auto lstore = [&](int buf) {
    unsigned char* st = lds + buf * STAGE;
    const float v[8] = {ga0.x, ga0.y, ga0.z, ga0.w, ga1.x, ga1.y, ga1.z, ga1.w};
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int e = 0; e < 8; e++) x3_split(v[e], h[e], m[e], l[e]);
    u32x4_t ph = {x3_pack(h[0], h[1]), x3_pack(h[2], h[3]), x3_pack(h[4], h[5]), x3_pack(h[6], h[7])};
    u32x4_t pm = {x3_pack(m[0], m[1]), x3_pack(m[2], m[3]), x3_pack(m[4], m[5]), x3_pack(m[6], m[7])};
    u32x4_t pl = {x3_pack(l[0], l[1]), x3_pack(l[2], l[3]), x3_pack(l[4], l[5]), x3_pack(l[6], l[7])};
    *reinterpret_cast<u32x4_t*>(st + 0 * PIECE + s_off) = ph;
    *reinterpret_cast<u32x4_t*>(st + 1 * PIECE + s_off) = pm;
    *reinterpret_cast<u32x4_t*>(st + 2 * PIECE + s_off) = pl;
    *reinterpret_cast<u32x4_t*>(st + 3 * PIECE + s_off) = gb0;
    *reinterpret_cast<u32x4_t*>(st + 4 * PIECE + s_off) = gb1;
    *reinterpret_cast<u32x4_t*>(st + 5 * PIECE + s_off) = gb2;
  };

  gload(0);
  lstore(0);
  gload(1);
  __syncthreads();
  for (int it = 0; it < NK; it++) {
    const int buf = it & 1;
    // tile it+1 (in registers since the previous iteration) -> the other stage; then fetch tile it+2
    if (it + 1 < NK) lstore(buf ^ 1);
    gload(it + 2);
    const unsigned char* st = lds + buf * STAGE;
    bf16x8_t A[2][3], B[2][3];
#pragma unroll
    for (int p = 0; p < 3; p++) {
#pragma unroll
      for (int i = 0; i < 2; i++) A[i][p] = *reinterpret_cast<const bf16x8_t*>(st + p * PIECE + fa[i]);
#pragma unroll
      for (int j = 0; j < 2; j++) B[j][p] = *reinterpret_cast<const bf16x8_t*>(st + (3 + p) * PIECE + fb[j]);
    }
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++) {
        // smallest terms first
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][2], B[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][2], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], B[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][1], B[j][0], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][1], acc[i][j], 0, 0, 0);
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i][0], B[j][0], acc[i][j], 0, 0, 0);
      }
    __syncthreads();
  }

  // --- epilogue (same as conv_tile): BN(scale,shift) + ReLU (+ dual add + ReLU), interior of padded NHWC
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
      if (DUAL) {
        int c = n_tile * (BNT / 2) + wn * 32 + (lane & 31);
        if (mvalid && c < a.Cout_p) {
          float4 e = reinterpret_cast<const float4*>(a.ep)[(size_t)p * a.Cout_p + c];
          float va = acc[i][0][r] * e.x + e.y;
          float vb = acc[i][1][r] * e.z + e.w;
          va = va > 0.f ? va : 0.f;
          vb = vb > 0.f ? vb : 0.f;
          float s = va + vb;
          a.y[obase + c] = s > 0.f ? s : 0.f;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 2; j++) {
          int c = n0 + (wn * 2 + j) * 32 + (lane & 31);
          if (mvalid && c < a.Cout_p) {
            if (a.raw) {
              a.y[obase + c] = acc[i][j][r];
            } else {
              float2 e = reinterpret_cast<const float2*>(a.ep)[(size_t)p * a.Cout_p + c];
              float v = acc[i][j][r] * e.x + e.y;
              a.y[obase + c] = v > 0.f ? v : 0.f;
            }
          }
        }
      }
    }
  }
}

