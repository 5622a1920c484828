// Dual network on gfx950: fp32 MFMA implicit-GEMM 3x3 conv tower + fused heads.
//
// Reference path replaced: dualnet/dual.go:50-103 (graph), dualnet/ermahagerdmonards.go:33-104 (layers),
// dualnet/meta.go:125-190 (Inferencer).  The reference evaluates ONE real board in row 0 of an
// ActionSpace-row batch per leaf (meta.go:128,175-177); here every row of the batch is a real leaf.
//
// Data layout in HBM
//   activations  padded NHWC fp32  [B][H+2][W+2][Cp]  (Cp = channels rounded up to 32; the 1-cell halo is
//                zero and never written, so the 3x3 taps need no bounds checks)
//   conv weights [tap 9][n][Cin_p] fp32, Cin contiguous (both GEMM operands are "rows of 32 contiguous k")
//                dual-branch blocks: n ordered per block tile as [a-channels | matching b-channels] so one
//                wave holds conv_a and conv_b of the same (pixel, channel) in matching MFMA accumulators
//   epilogue     per (position, channel) {scale, shift} (BN folded; gorgonia's gamma/beta are [B,C,H,W]-shaped,
//                row 0 is what inference uses — SURVEY App. B b3-b5)
//
// Kernel K1/K2: implicit GEMM  D[m, n] = sum_{tap, c} X[pix(m)+off(tap)][c] * Wt[tap][n][c]
//   M = B*H*W pixels, N = out channels (x2 for the dual block), K = 9*Cin
//   v_mfma_f32_32x32x2_f32 (exact fp32; 64 FLOP/clk/SIMD = 157.3 TF chip peak) — bound: MFMA.
//   256 threads = 4 waves; each wave owns MT x 2 tiles of 32x32; BK = 32; LDS double-buffered with a
//   16-byte-chunk XOR swizzle ((row>>1)&7) so the ds_read_b128 operand reads are bank-conflict free.
//   Algorithmic FLOPs per launch: 2 * M * N * 9 * Cin.
#include "net.hpp"
#include "gemm_maps.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace agz {

struct ConvArgs {
  const float* x;
  const float* w;
  const void* ep;
  float* y;
  int M, HW, W, Wp, HpWp;
  int Cin_p, Cout_p, Ntot;
  int n_mtiles, n_ntiles;
  int raw;  // 1: store the GEMM result as is (training forward / data-gradient passes), no BN/ReLU epilogue
  int splits;   // >1: split-K — block (tile, s) covers K-iterations [s*per, (s+1)*per) and stores raw partials to ws
  int per;
  float* ws;    // [splits][M][Ntot] partial sums (GEMM column order)
  // fp16x2 mode (conv_h2.hpp): amax_in[b] = max |activation| of board b in this layer's input (board_amax_kernel),
  // w_unscale = 2^-eb of the pre-scaled weights
  const unsigned* amax_in;
  float w_unscale;
  const unsigned* w_amax_dev;   // raw fp16x2 form (conv3x3_raw_h2): bits of max|w|, the scale of the device-built weight image
  unsigned* amax_out;           // conv3x3_x3_kernel<false>, != nullptr: atomicMax of the output's bits per board (zeroed by the caller): the
                                // range words board_amax_kernel would compute from the stored tensor
};

__device__ __forceinline__ int swz_off(int row, int chunk) { return row * 32 + ((chunk ^ ((row >> 1) & 7)) << 2); }

// One block tile: WM x WN waves, each wave MT x 2 MFMA tiles.  BM = WM*MT*32 rows (pixels), BNT = WN*64 GEMM columns.
template <int WM, int WN, int MT, bool DUAL>
__device__ __forceinline__ void conv_tile(const ConvArgs& a, float* lds, const int m0, const int n_tile, const int split) {
  constexpr int BM = WM * MT * 32;
  constexpr int BNT = WN * 64;
  static_assert(WM * WN == 4, "4 waves");
  constexpr int STAGE = (BM + BNT) * 32;  // floats per pipeline stage: A tile then B tile
  const int n0 = n_tile * BNT;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid / WN, wn = wid % WN;

  // ---- staging: global -> VGPR -> LDS, 16-byte chunks, XOR-swizzled LDS image -------------------------------------
  // (direct global->LDS DMA was measured slower here: its per-CU fill rate is too close to the 18 GB/s/CU this
  //  kernel streams)
  constexpr int A_PER_T = BM * 8 / 256;   // 16-byte chunks per thread per tile
  constexpr int B_PER_T = BNT * 8 / 256;
  static_assert((A_PER_T == 4 || A_PER_T == 2) && (B_PER_T == 4 || B_PER_T == 2), "staging layout");
  const int chunk = tid & 7;
  int a_goff[A_PER_T], a_loff[A_PER_T];
#pragma unroll
  for (int i = 0; i < A_PER_T; i++) {
    int row = (tid >> 3) + 32 * i;
    int m = m0 + row;
    if (m >= a.M) m = a.M - 1;
    int b = m / a.HW, p = m - b * a.HW;
    int h = p / a.W, w = p - h * a.W;
    a_goff[i] = ((b * a.HpWp) + (h + 1) * a.Wp + (w + 1)) * a.Cin_p + chunk * 4;
    a_loff[i] = swz_off(row, chunk);
  }
  int b_goff[B_PER_T], b_loff[B_PER_T];
#pragma unroll
  for (int i = 0; i < B_PER_T; i++) {
    int row = (tid >> 3) + 32 * i;
    int n = n0 + row;
    if (n >= a.Ntot) n = a.Ntot - 1;
    b_goff[i] = n * a.Cin_p + chunk * 4;
    b_loff[i] = swz_off(row, chunk);
  }
  const int NC = a.Cin_p >> 5;
  const int NKall = 9 * NC;
  const int it0 = split * a.per;                                   // this block's K-iteration range
  const int NK = (it0 + a.per < NKall) ? it0 + a.per : NKall;      // (exclusive end; `NK` keeps the loop macros unchanged)
  const int w_tap_stride = a.Ntot * a.Cin_p;

  // Staging registers are NAMED scalars: arrays get demoted to scratch/LDS by hipcc here (CDNA guide rule 20).
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;  // set 0
  float4 sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;  // set 1
  int xo_, wo_;
#define AGZ_TILE_OFFS(IT)                                                                   \
  {                                                                                         \
    int t_ = (IT) < NK ? (IT) : NK - 1;                                                     \
    int tap_ = t_ / NC, cc_ = t_ - tap_ * NC;                                               \
    int ky_ = tap_ / 3, kx_ = tap_ - ky_ * 3;                                               \
    xo_ = ((ky_ - 1) * a.Wp + (kx_ - 1)) * a.Cin_p + cc_ * 32;                              \
    wo_ = tap_ * w_tap_stride + cc_ * 32;                                                   \
  }
#define AGZ_GLA(R, I) if constexpr ((I) < A_PER_T) R = *reinterpret_cast<const float4*>(a.x + a_goff[(I) < A_PER_T ? (I) : 0] + xo_);
#define AGZ_GLB(R, I) if constexpr ((I) < B_PER_T) R = *reinterpret_cast<const float4*>(a.w + b_goff[(I) < B_PER_T ? (I) : 0] + wo_);
#define AGZ_LSA(R, I, BUF) if constexpr ((I) < A_PER_T) *reinterpret_cast<float4*>(lds + (BUF) * STAGE + a_loff[(I) < A_PER_T ? (I) : 0]) = R;
#define AGZ_LSB(R, I, BUF) if constexpr ((I) < B_PER_T) *reinterpret_cast<float4*>(lds + (BUF) * STAGE + BM * 32 + b_loff[(I) < B_PER_T ? (I) : 0]) = R;
#define AGZ_SB __builtin_amdgcn_sched_barrier(0);

  f32x16 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  // operand rows of this lane
  int a_row[MT], b_row[2];
#pragma unroll
  for (int i = 0; i < MT; i++) a_row[i] = (wm * MT + i) * 32 + (lane & 31);
  if (DUAL) {
    b_row[0] = wn * 32 + (lane & 31);            // branch a channels
    b_row[1] = WN * 32 + wn * 32 + (lane & 31);  // matching branch b channels
  } else {
    b_row[0] = (wn * 2 + 0) * 32 + (lane & 31);
    b_row[1] = (wn * 2 + 1) * 32 + (lane & 31);
  }
  const int khalf = lane >> 5;

  // Operand (fragment) reads: inline-asm ds_read_b128 into registers that stay live across the whole loop ("+v").
  // Letting hipcc allocate short-lived fragment registers makes it alias them onto staging registers whose global
  // loads are still in flight, and it then protects the overwrite with s_waitcnt vmcnt(0) at the top of every
  // iteration — the whole L2 latency exposed 72 times per tile (measured: -7%).  The asm reads carry hand-counted
  // lgkmcnt waits (LDS ops retire in order; the interleaved ds_writes are counted too), each followed by a
  // sched_barrier because hipcc may hoist register-only MFMAs above an asm wait (guide §5.4 rule 18).
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) float*)lds);
  unsigned a_addr[MT], b_addr[2], a_sw[MT], b_sw[2];
#pragma unroll
  for (int i = 0; i < MT; i++) { a_addr[i] = lds_base + a_row[i] * 128; a_sw[i] = (a_row[i] >> 1) & 7; }
#pragma unroll
  for (int j = 0; j < 2; j++) { b_addr[j] = lds_base + BM * 128 + b_row[j] * 128; b_sw[j] = (b_row[j] >> 1) & 7; }
  f32x4 f0a[MT], f0b[2], f1a[MT], f1b[2];  // native vectors: "+v" asm operands must not be HIP's struct float4
#pragma unroll
  for (int i = 0; i < MT; i++) { f0a[i] = f32x4{0.f, 0.f, 0.f, 0.f}; f1a[i] = f0a[i]; }
#pragma unroll
  for (int j = 0; j < 2; j++) { f0b[j] = f32x4{0.f, 0.f, 0.f, 0.f}; f1b[j] = f0b[j]; }
#define AGZ_LDSR(DST, ADDR) asm volatile("ds_read_b128 %0, %1" : "+v"(DST) : "v"(ADDR));
#define AGZ_FRAG_READ(AV, BV, KS, BUF)                                                                            \
  {                                                                                                               \
    _Pragma("unroll") for (int i = 0; i < MT; i++) {                                                              \
      unsigned ad_ = a_addr[i] + (BUF) * (STAGE * 4) + (((2 * (KS) + khalf) ^ a_sw[i]) << 4);                     \
      AGZ_LDSR(AV[i], ad_)                                                                                        \
    }                                                                                                             \
    _Pragma("unroll") for (int j = 0; j < 2; j++) {                                                               \
      unsigned ad_ = b_addr[j] + (BUF) * (STAGE * 4) + (((2 * (KS) + khalf) ^ b_sw[j]) << 4);                     \
      AGZ_LDSR(BV[j], ad_)                                                                                        \
    }                                                                                                             \
  }
#define AGZ_WAIT_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N)); __builtin_amdgcn_sched_barrier(0);
#define AGZ_MQ(AV, BV, C)                                                                                         \
  _Pragma("unroll") for (int i = 0; i < MT; i++) _Pragma("unroll") for (int j = 0; j < 2; j++)                    \
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(AV[i].C, BV[j].C, acc[i][j], 0, 0, 0);
  constexpr int NRD = MT + 2;               // ds_reads per fragment set
  constexpr int NWA = A_PER_T, NWB = B_PER_T;  // ds_writes in the two store phases

  // ---- main loop: 3-deep software pipeline ---------------------------------------------------------------------
  //   tile it   : multiplied out of LDS stage it&1
  //   tile it+1 : sits in staging set X (loaded during the previous iteration), written to stage (it+1)&1 now
  //   tile it+2 : global loads issued now into staging set Y
  // Every global_load / ds_write / ds_read is slotted BETWEEN MFMA quads and pinned with sched_barrier: a wave
  // issues in order, so a memory instruction placed behind an MFMA costs nothing while the 64-cycle MFMA occupies
  // the matrix pipe; clumped at the top/bottom of the iteration they idle it.
#define AGZ_ITER(IT, BUF, XA0, XA1, XA2, XA3, XB0, XB1, XB2, XB3, YA0, YA1, YA2, YA3, YB0, YB1, YB2, YB3)         \
  {                                                                                                               \
    AGZ_TILE_OFFS((IT) + 2)                                                                                       \
    AGZ_FRAG_READ(f0a, f0b, 0, BUF)                                                                               \
    AGZ_FRAG_READ(f1a, f1b, 1, BUF)                                                                               \
    AGZ_WAIT_LGKM(NRD) /* k-step 0 operands landed */                                                             \
    AGZ_MQ(f0a, f0b, x) AGZ_SB AGZ_GLA(YA0, 0) AGZ_SB AGZ_MQ(f0a, f0b, y) AGZ_SB AGZ_GLA(YA1, 1)                  \
    AGZ_SB AGZ_MQ(f0a, f0b, z) AGZ_SB AGZ_GLA(YA2, 2) AGZ_SB AGZ_MQ(f0a, f0b, w) AGZ_SB AGZ_GLA(YA3, 3)           \
    AGZ_SB AGZ_FRAG_READ(f0a, f0b, 2, BUF)                                                                        \
    AGZ_WAIT_LGKM(NRD) /* k-step 1 */                                                                             \
    AGZ_MQ(f1a, f1b, x) AGZ_SB AGZ_GLB(YB0, 0) AGZ_SB AGZ_MQ(f1a, f1b, y) AGZ_SB AGZ_GLB(YB1, 1)                  \
    AGZ_SB AGZ_MQ(f1a, f1b, z) AGZ_SB AGZ_GLB(YB2, 2) AGZ_SB AGZ_MQ(f1a, f1b, w) AGZ_SB AGZ_GLB(YB3, 3)           \
    AGZ_SB AGZ_FRAG_READ(f1a, f1b, 3, BUF)                                                                        \
    AGZ_WAIT_LGKM(NRD) /* k-step 2 */                                                                             \
    AGZ_MQ(f0a, f0b, x) AGZ_SB AGZ_LSA(XA0, 0, (BUF) ^ 1) AGZ_SB AGZ_MQ(f0a, f0b, y) AGZ_SB AGZ_LSA(XA1, 1, (BUF) ^ 1) \
    AGZ_SB AGZ_MQ(f0a, f0b, z) AGZ_SB AGZ_LSA(XA2, 2, (BUF) ^ 1) AGZ_SB AGZ_MQ(f0a, f0b, w) AGZ_SB AGZ_LSA(XA3, 3, (BUF) ^ 1) \
    AGZ_SB                                                                                                        \
    AGZ_WAIT_LGKM(NWA) /* k-step 3 operands: only the A-part ds_writes may still be outstanding */                \
    AGZ_MQ(f1a, f1b, x) AGZ_SB AGZ_LSB(XB0, 0, (BUF) ^ 1) AGZ_SB AGZ_MQ(f1a, f1b, y) AGZ_SB AGZ_LSB(XB1, 1, (BUF) ^ 1) \
    AGZ_SB AGZ_MQ(f1a, f1b, z) AGZ_SB AGZ_LSB(XB2, 2, (BUF) ^ 1) AGZ_SB AGZ_MQ(f1a, f1b, w) AGZ_SB AGZ_LSB(XB3, 3, (BUF) ^ 1) \
    AGZ_SB                                                                                                        \
    __syncthreads();                                                                                              \
  }
  (void)NWB;
  // prologue: tile 0 -> LDS stage 0 (through set 1), tile 1 -> set 0
  AGZ_TILE_OFFS(it0)
  AGZ_GLA(sa0, 0) AGZ_GLA(sa1, 1) AGZ_GLA(sa2, 2) AGZ_GLA(sa3, 3) AGZ_GLB(sb0, 0) AGZ_GLB(sb1, 1) AGZ_GLB(sb2, 2) AGZ_GLB(sb3, 3)
  AGZ_TILE_OFFS(it0 + 1)
  AGZ_GLA(ra0, 0) AGZ_GLA(ra1, 1) AGZ_GLA(ra2, 2) AGZ_GLA(ra3, 3) AGZ_GLB(rb0, 0) AGZ_GLB(rb1, 1) AGZ_GLB(rb2, 2) AGZ_GLB(rb3, 3)
  AGZ_LSA(sa0, 0, 0) AGZ_LSA(sa1, 1, 0) AGZ_LSA(sa2, 2, 0) AGZ_LSA(sa3, 3, 0) AGZ_LSB(sb0, 0, 0) AGZ_LSB(sb1, 1, 0) AGZ_LSB(sb2, 2, 0) AGZ_LSB(sb3, 3, 0)
  __syncthreads();
  for (int it = it0; it < NK; it += 2) {
    AGZ_ITER(it, 0, ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3)
    if (it + 1 < NK) AGZ_ITER(it + 1, 1, sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3, ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3)
  }
#undef AGZ_ITER
#undef AGZ_WAIT_LGKM
#undef AGZ_LDSR
#undef AGZ_MQ
#undef AGZ_FRAG_READ
#undef AGZ_SB
#undef AGZ_LSB
#undef AGZ_LSA
#undef AGZ_GLB
#undef AGZ_GLA
#undef AGZ_TILE_OFFS
  // C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  if (a.splits > 1) {  // split-K: raw partial sums, reduced + finished by splitk_epilogue_kernel
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        int m = m0 + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m >= a.M) continue;
#pragma unroll
        for (int j = 0; j < 2; j++) {
          int col = n0 + b_row[j];
          if (col < a.Ntot) a.ws[((size_t)split * a.M + m) * a.Ntot + col] = acc[i][j][r];
        }
      }
    return;
  }
  // --- epilogue: BN(scale,shift) + ReLU (+ dual add + ReLU), store interior of padded NHWC
#pragma unroll
  for (int i = 0; i < MT; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      int row = (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
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

template <int WM, int WN, int MT, bool DUAL>
__global__ __launch_bounds__(256, 2) void conv3x3_mfma_kernel(ConvArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * (WM * MT * 32 + WN * 64) * 32];
  // XCD-aware tile mapping (bijective): consecutive block ids land on different XCDs; give each XCD a contiguous
  // run of tiles so the n-tiles of one m-tile (same A rows) share an L2.
  const int nblk = a.n_mtiles * a.n_ntiles;
  const int id = blockIdx.x;
  int q = nblk >> 3, r = nblk & 7, xcd = id & 7, slot = id >> 3;
  int tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
  const int m_tile = tile / a.n_ntiles, n_tile = tile - m_tile * a.n_ntiles;
  conv_tile<WM, WN, MT, DUAL>(a, lds, m_tile * (WM * MT * 32), n_tile, blockIdx.y);
}

#include "conv_x3.hpp"
#include "conv_wino.hpp"
#include "conv_h2.hpp"
#include "conv_wino_h2.hpp"
#include "conv_wino_h2c.hpp"
#include "conv_lat.hpp"


// device-side weight split for the trainer (weights change every step): one thread per (tap, n, ci)
__global__ void split_w3_kernel(const float* __restrict__ w, unsigned short* __restrict__ w3, int N, int Cin_p) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)9 * N * Cin_p;
  if (idx >= total) return;
  int ci = (int)(idx % Cin_p);
  size_t r = idx / Cin_p;
  int n = (int)(r % N), t = (int)(r / N);
  unsigned h, m, l;
  x3_split(w[idx], h, m, l);
  size_t base = (((size_t)((ci >> 4) * 9 + t) * 3) * N + n) * 16 + (ci & 15);
  w3[base] = (unsigned short)(h >> 16);
  w3[base + (size_t)N * 16] = (unsigned short)(m >> 16);
  w3[base + (size_t)2 * N * 16] = (unsigned short)(l >> 16);
}

// split-K finish: sum the partials (in split order) and apply the conv epilogue.  One thread per (pixel row m, 4 channels).
template <bool DUAL>
__global__ __launch_bounds__(256) void splitk_epilogue_kernel(ConvArgs a, int half) {
  const int C4 = a.Cout_p >> 2;
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)a.M * C4) return;
  int m = (int)(idx / C4), c = ((int)(idx - (size_t)m * C4)) << 2;
  int b = m / a.HW, p = m - b * a.HW;
  int h = p / a.W, w = p - h * a.W;
  float4* out = reinterpret_cast<float4*>(a.y + ((size_t)b * a.HpWp + (h + 1) * a.Wp + (w + 1)) * a.Cout_p + c);
  const size_t sstride = (size_t)a.M * a.Ntot;
  if (DUAL) {
    int ca = (c / half) * 2 * half + (c % half);   // half is a multiple of 32: the 4 channels stay in one half
    const float* row = a.ws + (size_t)m * a.Ntot + ca;
    float4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 6
    for (int s = 0; s < a.splits; s++) {
      float4 va = *reinterpret_cast<const float4*>(row + s * sstride);
      float4 vb = *reinterpret_cast<const float4*>(row + s * sstride + half);
      sa.x += va.x; sa.y += va.y; sa.z += va.z; sa.w += va.w;
      sb.x += vb.x; sb.y += vb.y; sb.z += vb.z; sb.w += vb.w;
    }
    const float4* e = reinterpret_cast<const float4*>(a.ep) + (size_t)p * a.Cout_p + c;
    float r[4];
    const float av[4] = {sa.x, sa.y, sa.z, sa.w}, bv[4] = {sb.x, sb.y, sb.z, sb.w};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      float4 eq = e[q];
      float va = av[q] * eq.x + eq.y, vb = bv[q] * eq.z + eq.w;
      va = va > 0.f ? va : 0.f; vb = vb > 0.f ? vb : 0.f;
      float t = va + vb;
      r[q] = t > 0.f ? t : 0.f;
    }
    *out = float4{r[0], r[1], r[2], r[3]};
  } else {
    const float* row = a.ws + (size_t)m * a.Ntot + c;
    float4 sa = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 6
    for (int s = 0; s < a.splits; s++) {
      float4 va = *reinterpret_cast<const float4*>(row + s * sstride);
      sa.x += va.x; sa.y += va.y; sa.z += va.z; sa.w += va.w;
    }
    if (a.raw) { *out = sa; return; }
    const float2* e = reinterpret_cast<const float2*>(a.ep) + (size_t)p * a.Cout_p + c;
    const float av[4] = {sa.x, sa.y, sa.z, sa.w};
    float r[4];
#pragma unroll
    for (int q = 0; q < 4; q++) { float v = av[q] * e[q].x + e[q].y; r[q] = v > 0.f ? v : 0.f; }
    *out = float4{r[0], r[1], r[2], r[3]};
  }
}

// planes NCHW [B,F,H,W] -> padded NHWC [B][Hp][Wp][32] (channels >= F zero)
__global__ void pack_planes_kernel(const float* __restrict__ planes, float* __restrict__ out, int B, int F, int H, int W,
                                   int Fp) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;  // one thread per (b, h, w, c)
  int total = B * H * W * Fp;
  if (idx >= total) return;
  int c = idx % Fp;
  int p = idx / Fp;
  int w = p % W, h = (p / W) % H, b = p / (W * H);
  float v = c < F ? planes[((size_t)(b * F + c) * H + h) * W + w] : 0.f;
  out[(((size_t)b * (H + 2) + h + 1) * (W + 2) + w + 1) * Fp + c] = v;
}

struct HeadArgs {
  const float* x;  // padded NHWC [B][Hp][Wp][Kp]
  const float* conv;  // [3][Kp]
  const float* bn;    // [3][HW][2]
  const float* Wp; const float* bp; const float* W1; const float* b1; const float* W2; const float* b2;
  float* policy; float* value;
  int H, W, HW, Wp_, HpWp, Kp, A, FC;
  float* feat;  // latency regime: [B][3][HW]
  float* cols;  // latency regime: [B][A + FC] policy logits, then value hidden units
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// K4+K5: both heads for one board per workgroup (dual.go:72-97)
__global__ __launch_bounds__(256) void heads_kernel(HeadArgs a) {
  extern __shared__ float sm[];
  float* feat = sm;                  // [3][HW]  relu(bn(conv1x1)) : policy c0, policy c1, value
  float* red = sm + 3 * a.HW;        // [8]
  float* hid = red + 8;              // [FC]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  // phase 1: 1x1 convs — a wave per pixel, lanes over channels (coalesced 16 B per lane)
  for (int p = wid; p < a.HW; p += 4) {
    int h = p / a.W, w = p - h * a.W;
    const float* xp = a.x + ((size_t)b * a.HpWp + (h + 1) * a.Wp_ + (w + 1)) * a.Kp;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int c = lane * 4; c < a.Kp; c += 256) {
      float4 xv = *reinterpret_cast<const float4*>(xp + c);
      float4 w0 = *reinterpret_cast<const float4*>(a.conv + c);
      float4 w1 = *reinterpret_cast<const float4*>(a.conv + a.Kp + c);
      float4 w2 = *reinterpret_cast<const float4*>(a.conv + 2 * a.Kp + c);
      s0 += xv.x * w0.x + xv.y * w0.y + xv.z * w0.z + xv.w * w0.w;
      s1 += xv.x * w1.x + xv.y * w1.y + xv.z * w1.z + xv.w * w1.w;
      s2 += xv.x * w2.x + xv.y * w2.y + xv.z * w2.z + xv.w * w2.w;
    }
    s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
    if (lane == 0) {
      float v0 = s0 * a.bn[(0 * a.HW + p) * 2] + a.bn[(0 * a.HW + p) * 2 + 1];
      float v1 = s1 * a.bn[(1 * a.HW + p) * 2] + a.bn[(1 * a.HW + p) * 2 + 1];
      float v2 = s2 * a.bn[(2 * a.HW + p) * 2] + a.bn[(2 * a.HW + p) * 2 + 1];
      feat[p] = v0 > 0.f ? v0 : 0.f;
      feat[a.HW + p] = v1 > 0.f ? v1 : 0.f;
      feat[2 * a.HW + p] = v2 > 0.f ? v2 : 0.f;
    }
  }
  __syncthreads();
  // phase 2: policy logits = feat[0:2HW] . Wp[2HW, A] + bp ; softmax
  float lmax = -INFINITY;
  float logit[2];  // A <= 512 supported per 256 threads x 2
  for (int r = 0, j = tid; r < 2; r++, j += 256) {
    float s = 0.f;
    if (j < a.A) {
      for (int i = 0; i < 2 * a.HW; i++) s += feat[i] * a.Wp[(size_t)i * a.A + j];
      s += a.bp[j];
      lmax = fmaxf(lmax, s);
    }
    logit[r] = s;
  }
  for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
  if (lane == 0) red[wid] = lmax;
  __syncthreads();
  float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float lsum = 0.f;
  for (int r = 0, j = tid; r < 2; r++, j += 256) {
    if (j < a.A) { logit[r] = expf(logit[r] - mx); lsum += logit[r]; }
  }
  lsum = wave_sum(lsum);
  if (lane == 0) red[4 + wid] = lsum;
  __syncthreads();
  float tot = red[4] + red[5] + red[6] + red[7];
  for (int r = 0, j = tid; r < 2; r++, j += 256)
    if (j < a.A) a.policy[(size_t)b * a.A + j] = logit[r] / tot;
  // phase 3: value = tanh( relu(feat[2] . W1 + b1) . W2 + b2 )
  for (int j = tid; j < a.FC; j += 256) {
    float s = 0.f;
    const float* f2 = feat + 2 * a.HW;
    for (int i = 0; i < a.HW; i++) s += f2[i] * a.W1[(size_t)i * a.FC + j];
    s += a.b1[j];
    hid[j] = s > 0.f ? s : 0.f;
  }
  __syncthreads();
  float o = 0.f;
  for (int j = tid; j < a.FC; j += 256) o += hid[j] * a.W2[j];
  o = wave_sum(o);
  __syncthreads();
  if (lane == 0) red[wid] = o;
  __syncthreads();
  if (tid == 0) a.value[b] = tanhf(red[0] + red[1] + red[2] + red[3] + a.b2[0]);
}

// ---- latency regime heads (few boards): the same maths as heads_kernel spread over the chip in three launches ----
// (1) 1x1 convs + BN + ReLU: a wave per pixel
__global__ __launch_bounds__(256) void heads_feat_kernel(HeadArgs a) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, p = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= a.HW) return;
  int h = p / a.W, w = p - h * a.W;
  const float* xp = a.x + ((size_t)b * a.HpWp + (h + 1) * a.Wp_ + (w + 1)) * a.Kp;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int c = lane * 4; c < a.Kp; c += 256) {
    float4 xv = *reinterpret_cast<const float4*>(xp + c);
    float4 w0 = *reinterpret_cast<const float4*>(a.conv + c);
    float4 w1 = *reinterpret_cast<const float4*>(a.conv + a.Kp + c);
    float4 w2 = *reinterpret_cast<const float4*>(a.conv + 2 * a.Kp + c);
    s0 += xv.x * w0.x + xv.y * w0.y + xv.z * w0.z + xv.w * w0.w;
    s1 += xv.x * w1.x + xv.y * w1.y + xv.z * w1.z + xv.w * w1.w;
    s2 += xv.x * w2.x + xv.y * w2.y + xv.z * w2.z + xv.w * w2.w;
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) {
    float* f = a.feat + (size_t)b * 3 * a.HW;
    float v0 = s0 * a.bn[(0 * a.HW + p) * 2] + a.bn[(0 * a.HW + p) * 2 + 1];
    float v1 = s1 * a.bn[(1 * a.HW + p) * 2] + a.bn[(1 * a.HW + p) * 2 + 1];
    float v2 = s2 * a.bn[(2 * a.HW + p) * 2] + a.bn[(2 * a.HW + p) * 2 + 1];
    f[p] = v0 > 0.f ? v0 : 0.f;
    f[a.HW + p] = v1 > 0.f ? v1 : 0.f;
    f[2 * a.HW + p] = v2 > 0.f ? v2 : 0.f;
  }
}
// (2) both FC layers as one column space [A policy logits | FC value hidden units]: 64 columns per workgroup,
//     16 waves each own a slice of the reduction rows, partials combined in wave order through LDS
__global__ __launch_bounds__(1024) void heads_fc_kernel(HeadArgs a) {
  __shared__ float part[16][64];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const bool is_pol = col < a.A;
  const int j = is_pol ? col : col - a.A;
  const bool live = col < a.A + a.FC;
  const int rows = is_pol ? 2 * a.HW : a.HW, ld = is_pol ? a.A : a.FC;
  const float* Wm = is_pol ? a.Wp : a.W1;
  const float* f = a.feat + (size_t)b * 3 * a.HW + (is_pol ? 0 : 2 * a.HW);
  const int chunk = (rows + 15) >> 4;
  int i0 = wv * chunk, i1 = i0 + chunk < rows ? i0 + chunk : rows;
  float s = 0.f;
  if (live) {
    // sixteen rows' loads in flight at a time (the plain loop waited for every row's pair of loads: 45 dependent round trips per wave,
    // 22 us for a 1.8 MB matrix-vector product); the additions keep their order
    int i = i0;
    for (; i + 16 <= i1; i += 16) {
      float fv[16], wv_[16];
#pragma unroll
      for (int u = 0; u < 16; u++) { fv[u] = f[i + u]; wv_[u] = Wm[(size_t)(i + u) * ld + j]; }
#pragma unroll
      for (int u = 0; u < 16; u++) s = fmaf(fv[u], wv_[u], s);   // (explicit: both forms of this kernel must round alike)
    }
    {
      float fv[16], wv_[16];
#pragma unroll
      for (int u = 0; u < 16; u++) { const bool ok = i + u < i1; const int r = ok ? i + u : 0; fv[u] = ok ? f[r] : 0.f; wv_[u] = Wm[(size_t)r * ld + j]; }   // (row 0: always inside the matrix)
#pragma unroll
      for (int u = 0; u < 16; u++) if (i + u < i1) s = fmaf(fv[u], wv_[u], s);
    }
  }
  part[wv][lane] = s;
  __syncthreads();
  if (wv == 0 && live) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; q++) t += part[q][lane];
    t += is_pol ? a.bp[j] : a.b1[j];
    if (!is_pol) t = t > 0.f ? t : 0.f;
    a.cols[(size_t)b * (a.A + a.FC) + col] = t;
  }
}
// (2b) the same columns for NB boards per workgroup (many boards: every weight is read once per NB boards instead of once per board —
//      5120 workgroups x 185 KB of weights were 0.95 GB through L2 per 512-board pass, 87 us).  Per (board, column) the additions run in
//      exactly heads_fc_kernel's order (sixteen-row groups, masked tail, partials in wave order): bit-identical outputs.
template <int NB>
__global__ __launch_bounds__(1024) void heads_fc_nb_kernel(HeadArgs a, int B) {
  __shared__ float part[NB][16][64];
  const int b0 = blockIdx.y * NB, lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = blockIdx.x * 64 + lane;
  const bool is_pol = col < a.A;
  const int j = is_pol ? col : col - a.A;
  const bool live = col < a.A + a.FC;
  const int rows = is_pol ? 2 * a.HW : a.HW, ld = is_pol ? a.A : a.FC;
  const float* Wm = is_pol ? a.Wp : a.W1;
  const float* f[NB];
#pragma unroll
  for (int n = 0; n < NB; n++) f[n] = a.feat + (size_t)min(b0 + n, B - 1) * 3 * a.HW + (is_pol ? 0 : 2 * a.HW);
  const int chunk = (rows + 15) >> 4;
  int i0 = wv * chunk, i1 = i0 + chunk < rows ? i0 + chunk : rows;
  float s[NB];
#pragma unroll
  for (int n = 0; n < NB; n++) s[n] = 0.f;
  if (live) {
    int i = i0;
    for (; i + 16 <= i1; i += 16) {
      float wv_[16];
#pragma unroll
      for (int u = 0; u < 16; u++) wv_[u] = Wm[(size_t)(i + u) * ld + j];
#pragma unroll
      for (int n = 0; n < NB; n++) {
        float fv[16];
#pragma unroll
        for (int u = 0; u < 16; u++) fv[u] = f[n][i + u];
#pragma unroll
        for (int u = 0; u < 16; u++) s[n] = fmaf(fv[u], wv_[u], s[n]);
      }
    }
    {
      float wv_[16];
#pragma unroll
      for (int u = 0; u < 16; u++) { const int r = i + u < i1 ? i + u : 0; wv_[u] = Wm[(size_t)r * ld + j]; }   // (row 0: always inside the matrix)
#pragma unroll
      for (int n = 0; n < NB; n++) {
        float fv[16];
#pragma unroll
        for (int u = 0; u < 16; u++) { const bool ok = i + u < i1; fv[u] = ok ? f[n][ok ? i + u : 0] : 0.f; }
#pragma unroll
        for (int u = 0; u < 16; u++) if (i + u < i1) s[n] = fmaf(fv[u], wv_[u], s[n]);
      }
    }
  }
#pragma unroll
  for (int n = 0; n < NB; n++) part[n][wv][lane] = s[n];
  __syncthreads();
  if (live && wv < NB && b0 + wv < B) {             // wave n finishes board n
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; q++) t += part[wv][q][lane];
    t += is_pol ? a.bp[j] : a.b1[j];
    if (!is_pol) t = t > 0.f ? t : 0.f;
    a.cols[(size_t)(b0 + wv) * (a.A + a.FC) + col] = t;
  }
}
// (3) softmax over the logits; value = tanh(hidden . W2 + b2)
__global__ __launch_bounds__(256) void heads_out_kernel(HeadArgs a) {
  __shared__ float red[8];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const float* cols = a.cols + (size_t)b * (a.A + a.FC);
  float lmax = -INFINITY;
  float logit[2];
  for (int r = 0, j = tid; r < 2; r++, j += 256) {
    logit[r] = j < a.A ? cols[j] : 0.f;
    if (j < a.A) lmax = fmaxf(lmax, logit[r]);
  }
  for (int o = 32; o > 0; o >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
  if (lane == 0) red[wid] = lmax;
  __syncthreads();
  float mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float lsum = 0.f;
  for (int r = 0, j = tid; r < 2; r++, j += 256)
    if (j < a.A) { logit[r] = expf(logit[r] - mx); lsum += logit[r]; }
  lsum = wave_sum(lsum);
  if (lane == 0) red[4 + wid] = lsum;
  __syncthreads();
  float tot = red[4] + red[5] + red[6] + red[7];
  for (int r = 0, j = tid; r < 2; r++, j += 256)
    if (j < a.A) a.policy[(size_t)b * a.A + j] = logit[r] / tot;
  float o = 0.f;
  for (int j = tid; j < a.FC; j += 256) o += cols[a.A + j] * a.W2[j];
  o = wave_sum(o);
  __syncthreads();
  if (lane == 0) red[wid] = o;
  __syncthreads();
  if (tid == 0) a.value[b] = tanhf(red[0] + red[1] + red[2] + red[3] + a.b2[0]);
}

}  // namespace agz

using namespace agz;

// ------------------------------------------------------------------------------------------------
void agz_net::free_device() {
  auto f = [](float*& p) { if (p) { hipFree(p); p = nullptr; } };
  f(d_w_init); f(d_w_init_t); f(d_ep_init);
  if (d_w3_init) { hipFree(d_w3_init); d_w3_init = nullptr; }
  for (auto& p : d_w_dual) f(p);
  for (auto& p : d_ep_dual) f(p);
  for (auto& p : d_w3_dual) if (p) { hipFree(p); p = nullptr; }
  for (auto& p : d_w2_dual) if (p) { hipFree(p); p = nullptr; }
  if (d_amax) { hipFree(d_amax); d_amax = nullptr; }
  amax_cap = 0;
  for (auto& p : d_u3_dual) if (p) { hipFree(p); p = nullptr; }
  for (auto& p : d_u2_dual) if (p) { hipFree(p); p = nullptr; }
  d_u2_dual.clear();
  for (auto& p : d_u2_tin) if (p) hipFree(p);
  d_u2_tin.clear();
  for (auto& p : d_u2_colun) if (p) hipFree(p);
  d_u2_colun.clear();
  free_u2c();
  for (auto& p : d_ep_h2) if (p) hipFree(p);
  d_ep_h2.clear();
  if (d_ep_init_h2) { hipFree(d_ep_init_h2); d_ep_init_h2 = nullptr; }
  f(d_wV); f(d_wM);
  wino_chunk_cap = 0; wino_v_cap = 0;
  d_w_dual.clear(); d_ep_dual.clear(); d_w3_dual.clear(); d_w2_dual.clear(); d_u3_dual.clear();
  f(d_head_conv); f(d_head_bn); f(d_Wp); f(d_bp); f(d_W1); f(d_b1); f(d_W2); f(d_b2);
  f(d_act_in); f(d_actA); f(d_actB); f(d_planes); f(d_policy); f(d_value); f(d_ws); f(d_hs);
  ws_cap = 0; hs_cap = 0;
  max_batch = 0;
}

int agz_net::ensure_batch(int B) {
  if (B <= max_batch) return AGZ_OK;
  // the conv kernels index activations with 32-bit element offsets (and GEMM rows with int)
  AGZ_REQUIRE((size_t)B * Hp * Wp * (size_t)std::max(Kp, Fp) < ((size_t)1 << 31), AGZ_E_UNSUPPORTED,
              "agz_net: batch %d too large for 32-bit activation offsets (%d x %d board, %d channels)", B, H, W, Kp);
  auto f = [](float*& p) { if (p) { hipFree(p); p = nullptr; } };
  f(d_act_in); f(d_actA); f(d_actB); f(d_planes); f(d_policy); f(d_value);
  size_t px = (size_t)B * Hp * Wp;
  AGZ_HIP_TRY(hipMalloc(&d_act_in, px * Fp * sizeof(float)));
  AGZ_HIP_TRY(hipMalloc(&d_actA, px * Kp * sizeof(float)));
  AGZ_HIP_TRY(hipMalloc(&d_actB, px * Kp * sizeof(float)));
  AGZ_HIP_TRY(hipMalloc(&d_planes, (size_t)B * conf.Features * HW * sizeof(float)));
  AGZ_HIP_TRY(hipMalloc(&d_policy, (size_t)B * conf.ActionSpace * sizeof(float)));
  AGZ_HIP_TRY(hipMalloc(&d_value, (size_t)B * sizeof(float)));
  // zero halos once; kernels only ever write the interior
  AGZ_HIP_TRY(hipMemsetAsync(d_act_in, 0, px * Fp * sizeof(float), ctx->stream));
  AGZ_HIP_TRY(hipMemsetAsync(d_actA, 0, px * Kp * sizeof(float), ctx->stream));
  AGZ_HIP_TRY(hipMemsetAsync(d_actB, 0, px * Kp * sizeof(float), ctx->stream));
  max_batch = B;
  return AGZ_OK;
}

// Small batches (tournament-style Agent.Search with one tree = batch 1: 361 rows -> 12 tiles on 256 CUs) get split-K:
// the 9*Cin/32 K-iterations of a tile are shared out over `splits` workgroups of `per` iterations each, partial sums
// go to a workspace and a second kernel reduces them (in split order) and applies the epilogue.  `per` (measured at
// K=256, batch 1: 4 -> 1.19, 8 -> 1.05, 12 -> 1.15 ms/simulation) depends on the layer shape only, never on the batch, so within the split regime results are bit-identical for every batch size.
// ws/ws_cap: caller-owned workspace (grown on demand); ws == nullptr: never split.
static int splitk_per(int NC) { return NC; }  // one filter tap (NC = Cin/32 iterations) per workgroup -> 9 splits
template <int WM, int WN, int MT, bool DUAL>
static int launch_conv(agz_ctx* ctx, ConvArgs& a, float** ws = nullptr, size_t* ws_cap = nullptr) {
  const int klass = DUAL ? AGZ_PROF_CONV : AGZ_PROF_CONV_INIT;
  constexpr int BM = WM * MT * 32, BNT = WN * 64;
  a.n_ntiles = ceil_div(a.Ntot, BNT);
  a.n_mtiles = ceil_div(a.M, BM);
  const int tiles = a.n_mtiles * a.n_ntiles;
  const int NK = 9 * (a.Cin_p >> 5);
  a.splits = 1; a.per = NK; a.ws = nullptr;
  if (ws && NK > splitk_per(a.Cin_p >> 5)) {
    a.per = splitk_per(a.Cin_p >> 5);
    a.splits = ceil_div(NK, a.per);
    size_t need = (size_t)a.splits * a.M * a.Ntot;
    if (need > *ws_cap) {
      AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
      if (*ws) hipFree(*ws);
      *ws = nullptr; *ws_cap = 0;
      AGZ_HIP_TRY(hipMalloc(ws, need * sizeof(float)));
      *ws_cap = need;
    }
    a.ws = *ws;
  }
  ProfScope ps(ctx, klass);
  dim3 grid(tiles, a.splits), block(256);
  hipLaunchKernelGGL((conv3x3_mfma_kernel<WM, WN, MT, DUAL>), grid, block, 0, ctx->stream, a);
  if (a.splits > 1) {
    size_t n = (size_t)a.M * (a.Cout_p / 4);
    hipLaunchKernelGGL((splitk_epilogue_kernel<DUAL>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, a, BNT / 2);
  }
  return AGZ_OK;
}

// Raw 3x3 convolution (no epilogue) on padded-NHWC tensors: y[pix][n] = sum_{tap,c} x[pix+off(tap)][c] * w[tap][n][c].
// Used by the trainer for the training-mode forward, and — with tap-flipped, transposed weights — for the data gradient.
int agz::conv3x3_raw(agz_ctx* ctx, const float* x, const float* w, float* y, int B, int H, int W, int Cin_p, int Cout_p) {
  ConvArgs a{};
  a.M = B * H * W; a.HW = H * W; a.W = W; a.Wp = W + 2; a.HpWp = (H + 2) * (W + 2);
  a.x = x; a.w = w; a.ep = nullptr; a.y = y; a.Cin_p = Cin_p; a.Cout_p = Cout_p; a.Ntot = Cout_p; a.raw = 1;
  if (Cout_p % 128 == 0) launch_conv<2, 2, 2, false>(ctx, a); else launch_conv<4, 1, 1, false>(ctx, a);
  AGZ_HIP_TRY(hipGetLastError());
  return AGZ_OK;
}

int agz::split_w3(agz_ctx* ctx, const float* w, unsigned short* w3, int N, int Cin_p) {
  size_t total = (size_t)9 * N * Cin_p;
  hipLaunchKernelGGL(split_w3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, w, w3, N, Cin_p);
  AGZ_HIP_TRY(hipGetLastError());
  return AGZ_OK;
}

int agz::conv3x3_raw_x3(agz_ctx* ctx, const float* x, const unsigned short* w3, float* y, int B, int H, int W, int Cin_p, int Cout_p) {
  AGZ_REQUIRE(Cin_p % 16 == 0, AGZ_E_INVALID, "conv3x3_raw_x3: Cin %d not a multiple of 16", Cin_p);
  AGZ_REQUIRE((size_t)B * (H + 2) * (W + 2) * Cin_p * sizeof(float) < ((size_t)1 << 32), AGZ_E_UNSUPPORTED, "conv3x3_raw_x3: tensor above 4 GiB");
  ConvArgs a{};
  a.M = B * H * W; a.HW = H * W; a.W = W; a.Wp = W + 2; a.HpWp = (H + 2) * (W + 2);
  a.x = x; a.w = nullptr; a.ep = nullptr; a.y = y; a.Cin_p = Cin_p; a.Cout_p = Cout_p; a.Ntot = Cout_p; a.raw = 1;
  a.n_ntiles = ceil_div(Cout_p, 128); a.n_mtiles = ceil_div(a.M, 128); a.splits = 1;
  ProfScope ps(ctx, AGZ_PROF_CONV_INIT);
  hipLaunchKernelGGL((conv3x3_x3_kernel<false>), dim3(a.n_mtiles * a.n_ntiles), dim3(256), 0, ctx->stream, a, w3);
  AGZ_HIP_TRY(hipGetLastError());
  return AGZ_OK;
}

// ---- raw convolution with fp16x2 products (the trainer's forward in AGZ_COMPUTE_WINO_H2): conv_h2.hpp's 128x256 kernel with a raw
// store, on a weight image built on the device from the current filter (range word + hi/lo split; the weights change every step)
__global__ __launch_bounds__(256) void w_absmax_kernel(const float* __restrict__ w, size_t n, unsigned* __restrict__ out_bits) {
  __shared__ unsigned sm[4];
  unsigned m = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned a = __float_as_uint(w[i]) & 0x7fffffffu;
    m = a > m ? a : m;
  }
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o, 64); m = t > m ? t : m; }
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned a = sm[0] > sm[1] ? sm[0] : sm[1], b = sm[2] > sm[3] ? sm[2] : sm[3];
    atomicMax(out_bits, a > b ? a : b);
  }
}
// (also clears what the NEXT kernels accumulate into: the board range words of this call and the weight range word of the next call —
// two words used alternately, so no memset launches between the kernels)
__global__ __launch_bounds__(256) void split_w2_kernel(const float* __restrict__ w, _Float16* __restrict__ w2, int N, int Cin_p, const unsigned* __restrict__ wmax,
                                                       unsigned* __restrict__ board_words, int B, unsigned* __restrict__ wmax_next) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.x == 0) {
    for (int b = threadIdx.x; b < B; b += 256) board_words[b] = 0u;
    if (threadIdx.x == 0) *wmax_next = 0u;
  }
  if (idx >= (size_t)9 * N * Cin_p) return;
  const int ci = (int)(idx % Cin_p);
  const size_t r = idx / Cin_p;
  const int n = (int)(r % N), t = (int)(r / N);
  float s, inv;
  h2_scales(*wmax, &s, &inv);
  const float xs = w[idx] * s;
  const _Float16 hi = (_Float16)xs;
  const _Float16 lo = (_Float16)(xs - (float)hi);
  const size_t base = (((size_t)((ci >> 5) * 9 + t) * 2) * N + n) * 32 + (ci & 31);
  w2[base] = hi;
  w2[base + (size_t)N * 32] = lo;
}
bool agz::conv3x3_raw_h2_fits(int B, int H, int W, int Cin_p, int Cout_p) {
  return Cin_p % 32 == 0 && Cout_p % 256 == 0 && (size_t)B * (H + 2) * (W + 2) * (size_t)std::max(Cin_p, Cout_p) * sizeof(float) < ((size_t)1 << 32);
}
// max|w| word + fp16x2 weight image of one layer (also clears the B board-range words of sc for a range sweep that may follow)
static int raw_h2_weights(agz_ctx* ctx, const float* w, int B, int Cin_p, int Cout_p, WinoRawScratch* sc, unsigned** wmax_out) {
  hipStream_t s = ctx->stream;
  const size_t w_elems = (size_t)9 * Cout_p * Cin_p;
  if (sc->w2_cap < w_elems * 2) {
    if (sc->w2) hipFree(sc->w2);
    sc->w2 = nullptr; sc->w2_cap = 0;
    AGZ_HIP_TRY(hipMalloc(&sc->w2, w_elems * 2 * sizeof(_Float16)));
    sc->w2_cap = w_elems * 2;
  }
  if (sc->h2_b_cap < B) {
    if (sc->h2_words) hipFree(sc->h2_words);
    sc->h2_words = nullptr; sc->h2_b_cap = 0;
    AGZ_HIP_TRY(hipMalloc(&sc->h2_words, ((size_t)B + 2) * sizeof(unsigned)));
    AGZ_HIP_TRY(hipMemsetAsync(sc->h2_words, 0, ((size_t)B + 2) * sizeof(unsigned), s));
    sc->h2_b_cap = B; sc->h2_flip = 0;
  }
  unsigned* wmax = sc->h2_words + sc->h2_b_cap + sc->h2_flip;          // zero: cleared at allocation / by the previous call's split kernel
  unsigned* wmax_next = sc->h2_words + sc->h2_b_cap + (sc->h2_flip ^ 1);
  sc->h2_flip ^= 1;
  hipLaunchKernelGGL(w_absmax_kernel, dim3((unsigned)std::min<size_t>((w_elems + 255) / 256, 256)), dim3(256), 0, s, w, w_elems, wmax);
  hipLaunchKernelGGL(split_w2_kernel, dim3((unsigned)((w_elems + 255) / 256)), dim3(256), 0, s, w, (_Float16*)sc->w2, Cout_p, Cin_p, wmax, sc->h2_words, B, wmax_next);
  *wmax_out = wmax;
  return AGZ_OK;
}
int agz::conv3x3_raw_h2_weights(agz_ctx* ctx, const float* w, int Cin_p, int Cout_p, WinoRawScratch* sc, const void** w2, const unsigned** w_amax) {
  unsigned* wmax = nullptr;
  int r = raw_h2_weights(ctx, w, std::max(sc->h2_b_cap, 1), Cin_p, Cout_p, sc, &wmax);
  if (r != AGZ_OK) return r;
  *w2 = sc->w2; *w_amax = wmax;
  AGZ_HIP_TRY(hipGetLastError());
  return AGZ_OK;
}
int agz::conv3x3_raw_h2(agz_ctx* ctx, const float* x, const float* w, float* y, int B, int H, int W, int Cin_p, int Cout_p, WinoRawScratch* sc,
                        const unsigned* ranges) {
  AGZ_REQUIRE(conv3x3_raw_h2_fits(B, H, W, Cin_p, Cout_p), AGZ_E_UNSUPPORTED, "conv3x3_raw_h2: shape %d x %dx%d, %d -> %d not supported", B, H, W, Cin_p, Cout_p);
  hipStream_t s = ctx->stream;
  unsigned* wmax = nullptr;
  { int r = raw_h2_weights(ctx, w, B, Cin_p, Cout_p, sc, &wmax); if (r != AGZ_OK) return r; }
  if (!ranges) hipLaunchKernelGGL(board_amax_parts_kernel, dim3(B * 8), dim3(256), 0, s, x, sc->h2_words, H * W, W, W + 2, (H + 2) * (W + 2), Cin_p, 8);
  ConvArgs a{};
  a.M = B * H * W; a.HW = H * W; a.W = W; a.Wp = W + 2; a.HpWp = (H + 2) * (W + 2);
  a.x = x; a.w = nullptr; a.ep = nullptr; a.y = y; a.Cin_p = Cin_p; a.Cout_p = Cout_p; a.Ntot = Cout_p; a.raw = 1;
  a.n_ntiles = Cout_p / 256; a.n_mtiles = ceil_div(a.M, 128); a.splits = 1;
  a.amax_in = ranges ? ranges : sc->h2_words; a.w_unscale = 1.f; a.w_amax_dev = wmax;
  ProfScope ps(ctx, AGZ_PROF_CONV_INIT);
  // (Round 5, measured and dropped: the same kernel on 256 x 256 tiles — 512 threads, two 64 KB stages, one barrier per K step, half the
  // weight re-reads through L2 -> CU (3.4 instead of 5 GB per G19 layer): 0.665 against 0.67 ms.  The kernel is bound by the split of its
  // activations — VALU and LDS-write issue per staged element — not by operand bytes.)
  hipLaunchKernelGGL(conv3x3_h2w_kernel, dim3(a.n_mtiles * a.n_ntiles), dim3(256), 0, s, a, (const _Float16*)sc->w2);
  AGZ_HIP_TRY(hipGetLastError());
  return AGZ_OK;
}

bool agz::conv3x3_raw_wino_h2_fits(int B, int H, int W, int Cin_p, int Cout_p) {
  const int tm = wino_h2_pick_tm(H, W), npos = (tm + 2) * (tm + 2);
  const size_t tiles = (size_t)B * ceil_div(H, tm) * ceil_div(W, tm);
  return Cin_p % 32 == 0 && Cin_p >= 32 && Cout_p % 32 == 0 &&
         wino_h2_rows(npos, tiles) * Cin_p * 4 < ((size_t)1 << 32) &&                 // 32-bit byte offsets into V
         (size_t)npos * (Cin_p / 32) * 2 * Cout_p * 64 < ((size_t)1 << 32) &&         // ... and into the weight image
         wino_h2_rows(npos, tiles) * Cout_p * 4 < ((size_t)1 << 32);                  // scalar position offsets of the output stage
}

void agz::wino_raw_scratch_free(WinoRawScratch* sc) {
  if (sc->V) hipFree(sc->V);
  if (sc->M) hipFree(sc->M);
  if (sc->U2) hipFree(sc->U2);
  if (sc->words) hipFree(sc->words);
  if (sc->w2) hipFree(sc->w2);
  if (sc->h2_words) hipFree(sc->h2_words);
  *sc = WinoRawScratch{};
}

static int raw_weights_reserve(RawWeights* rw, size_t bytes) {
  if (rw->cap >= bytes && rw->words) return AGZ_OK;
  raw_weights_free(rw);
  AGZ_HIP_TRY(hipMalloc(&rw->img, bytes));
  AGZ_HIP_TRY(hipMalloc(&rw->words, 4 * sizeof(unsigned)));
  rw->cap = bytes;
  return AGZ_OK;
}
void agz::raw_weights_free(RawWeights* rw) {
  if (rw->img) hipFree(rw->img);
  if (rw->words) hipFree(rw->words);
  *rw = RawWeights{};
}
int agz::conv3x3_raw_h2_weights_to(agz_ctx* ctx, hipStream_t st, const float* w, int Cin_p, int Cout_p, RawWeights* out) {
  (void)ctx;
  const size_t w_elems = (size_t)9 * Cout_p * Cin_p;
  int r = raw_weights_reserve(out, w_elems * 2 * sizeof(_Float16));
  if (r != AGZ_OK) return r;
  AGZ_HIP_TRY(hipMemsetAsync(out->words, 0, 4 * sizeof(unsigned), st));
  hipLaunchKernelGGL(w_absmax_kernel, dim3((unsigned)std::min<size_t>((w_elems + 255) / 256, 256)), dim3(256), 0, st, w, w_elems, out->words);
  hipLaunchKernelGGL(split_w2_kernel, dim3((unsigned)((w_elems + 255) / 256)), dim3(256), 0, st, w, (_Float16*)out->img, Cout_p, Cin_p, out->words, out->words + 2, 1,
                     out->words + 3);
  AGZ_HIP_TRY(hipGetLastError());
  return AGZ_OK;
}
int agz::conv3x3_raw_wino_h2_weights_to(agz_ctx* ctx, hipStream_t st, const float* w, int H, int W, int Cin_p, int Cout_p, RawWeights* out) {
  (void)ctx;
  const int tm = wino_h2_pick_tm(H, W), npos = (tm + 2) * (tm + 2);
  const size_t u_need = (size_t)npos * (Cin_p / 32) * 2 * Cout_p * 32;   // fp16 elements
  int r = raw_weights_reserve(out, u_need * sizeof(_Float16));
  if (r != AGZ_OK) return r;
  AGZ_HIP_TRY(hipMemsetAsync(out->words, 0, 4 * sizeof(unsigned), st));
  const unsigned gw = (unsigned)(((size_t)Cout_p * Cin_p + 255) / 256);
  float* unscale = reinterpret_cast<float*>(out->words + 1);
  if (tm == 5) {
    hipLaunchKernelGGL(wino_u_absmax_kernel<5>, dim3(gw), dim3(256), 0, st, w, Cout_p, Cin_p, out->words);
    hipLaunchKernelGGL(wino_u_build_kernel<5>, dim3(gw), dim3(256), 0, st, w, Cout_p, Cin_p, out->words, (_Float16*)out->img, unscale);
  } else {
    hipLaunchKernelGGL(wino_u_absmax_kernel<4>, dim3(gw), dim3(256), 0, st, w, Cout_p, Cin_p, out->words);
    hipLaunchKernelGGL(wino_u_build_kernel<4>, dim3(gw), dim3(256), 0, st, w, Cout_p, Cin_p, out->words, (_Float16*)out->img, unscale);
  }
  AGZ_HIP_TRY(hipGetLastError());
  return AGZ_OK;
}

int agz::conv3x3_raw_wino_h2(agz_ctx* ctx, const float* x, const float* w, float* y, int B, int H, int W, int Cin_p, int Cout_p, WinoRawScratch* sc,
                             const unsigned* ranges, const RawWeights* pre) {
  AGZ_REQUIRE(conv3x3_raw_wino_h2_fits(B, H, W, Cin_p, Cout_p), AGZ_E_UNSUPPORTED, "conv3x3_raw_wino_h2: shape %d x %dx%d, %d -> %d not supported", B, H, W, Cin_p, Cout_p);
  hipStream_t s = ctx->stream;
  const int tm = wino_h2_pick_tm(H, W), npos = (tm + 2) * (tm + 2);
  const size_t tiles = (size_t)B * ceil_div(H, tm) * ceil_div(W, tm);
  const int row_pad = 0;   // (rows of padding after every position's 128 tile rows: 1 / 4 / 16 measured no different from 0 at C = 512)
  const size_t v_need = wino_h2_rows(npos, tiles, row_pad) * Cin_p, m_need = wino_h2_rows(npos, tiles, row_pad) * Cout_p;
  const size_t u_need = pre ? 0 : (size_t)npos * (Cin_p / 32) * 2 * Cout_p * 32;   // fp16 elements
  if (v_need > sc->v_cap || m_need > sc->m_cap || u_need > sc->u_cap || B > sc->b_cap) {
    AGZ_HIP_TRY(hipStreamSynchronize(s));
    if (v_need > sc->v_cap) { if (sc->V) hipFree(sc->V); sc->V = nullptr; sc->v_cap = 0; AGZ_HIP_TRY(hipMalloc(&sc->V, v_need * 4)); sc->v_cap = v_need; }
    if (m_need > sc->m_cap) { if (sc->M) hipFree(sc->M); sc->M = nullptr; sc->m_cap = 0; AGZ_HIP_TRY(hipMalloc(&sc->M, m_need * 4)); sc->m_cap = m_need; }
    if (u_need > sc->u_cap) { if (sc->U2) hipFree(sc->U2); sc->U2 = nullptr; sc->u_cap = 0; AGZ_HIP_TRY(hipMalloc(&sc->U2, u_need * 2)); sc->u_cap = u_need; }
    if (B > sc->b_cap) { if (sc->words) hipFree(sc->words); sc->words = nullptr; sc->b_cap = 0; AGZ_HIP_TRY(hipMalloc(&sc->words, ((size_t)B + 2) * 4)); sc->b_cap = B; }
  }
  unsigned* umax = sc->words + sc->b_cap;
  const float* unscale = pre ? reinterpret_cast<const float*>(pre->words + 1) : reinterpret_cast<const float*>(sc->words + sc->b_cap + 1);
  const int Hp = H + 2, Wp = W + 2;
  // the layer's Winograd-domain weights and their scale
  if (!pre) {
    float* un = reinterpret_cast<float*>(sc->words + sc->b_cap + 1);
    AGZ_HIP_TRY(hipMemsetAsync(umax, 0, 4, s));
    const unsigned gw = (unsigned)(((size_t)Cout_p * Cin_p + 255) / 256);
    if (tm == 5) {
      hipLaunchKernelGGL(wino_u_absmax_kernel<5>, dim3(gw), dim3(256), 0, s, w, Cout_p, Cin_p, umax);
      hipLaunchKernelGGL(wino_u_build_kernel<5>, dim3(gw), dim3(256), 0, s, w, Cout_p, Cin_p, umax, (_Float16*)sc->U2, un);
    } else {
      hipLaunchKernelGGL(wino_u_absmax_kernel<4>, dim3(gw), dim3(256), 0, s, w, Cout_p, Cin_p, umax);
      hipLaunchKernelGGL(wino_u_build_kernel<4>, dim3(gw), dim3(256), 0, s, w, Cout_p, Cin_p, umax, (_Float16*)sc->U2, un);
    }
  }
  // per-board range of the input (training activations are signed: the kernel takes |x|)
  if (!ranges) {
    AGZ_HIP_TRY(hipMemsetAsync(sc->words, 0, (size_t)B * sizeof(unsigned), s));
    hipLaunchKernelGGL(board_amax_parts_kernel, dim3(B * 8), dim3(256), 0, s, x, sc->words, H * W, W, Wp, Hp * Wp, Cin_p, 8);
  }
  WinoH2Args hh{};
  WinoArgs& wa = hh.w;
  wa.x = x; wa.y = y; wa.V = sc->V; wa.Mb = sc->M; wa.ep = nullptr;
  wa.B = B; wa.H = H; wa.W = W; wa.Hp = Hp; wa.Wp = Wp; wa.C = Cin_p; wa.Cout_p = Cout_p; wa.Ntot = Cout_p;
  hh.U2 = (const _Float16*)(pre ? pre->img : sc->U2); hh.w_unscale = 1.f; hh.w_unscale_dev = unscale; hh.raw = 1; hh.tm = tm;
  hh.amax_in = ranges ? ranges : sc->words; hh.amax_out = nullptr; hh.wave_max = nullptr; hh.fuse_prev = 0; hh.row_pad = row_pad;
  wino_h2_launch(ctx, hh, Cout_p % 256 == 0, s);
  AGZ_HIP_TRY(hipGetLastError());
  return AGZ_OK;
}

// Winograd-domain weights of every dual block (conv_wino.hpp): columns [0,Kp) branch a, [Kp,2Kp) branch b, natural order
int agz_net::build_wino_weights() {
  AGZ_REQUIRE(cfg == 0, AGZ_E_UNSUPPORTED, "agz_net: the Winograd path needs K a multiple of 64");
  // wino_gemm_kernel addresses the weights of one block with 32-bit byte offsets
  AGZ_REQUIRE((size_t)36 * (Kp / 16) * 3 * (2 * Kp) * 32 < ((size_t)1 << 32), AGZ_E_UNSUPPORTED, "agz_net: K %d too wide for the Winograd weight image", conf.K);
  AGZ_HIP_TRY(hipSetDevice(ctx->device));
  for (auto& p : d_u3_dual) if (p) hipFree(p);
  d_u3_dual.assign(conf.SharedLayers, nullptr);
  const int K = conf.K;
  size_t pi = 3;   // parameters: init conv (w, gamma, beta), then per block a: (w, gamma, beta), b: (w, gamma, beta)
  std::vector<unsigned short> u3;
  for (int l = 0; l < conf.SharedLayers; l++, pi += 6) {
    const std::vector<float>& wa = params[pi].v;
    const std::vector<float>& wb = params[pi + 3].v;
    const int Kp_ = Kp;
    agz::wino_build_u3(u3, 2 * Kp, Kp, [&](int n, int ci, int tap) -> double {
      const int o = n < Kp_ ? n : n - Kp_;
      if (o >= K || ci >= K) return 0.0;
      return (double)(n < Kp_ ? wa : wb)[((size_t)o * K + ci) * 9 + tap];
    });
    AGZ_HIP_TRY(hipMalloc(&d_u3_dual[l], u3.size() * 2));
    AGZ_HIP_TRY(hipMemcpyAsync(d_u3_dual[l], u3.data(), u3.size() * 2, hipMemcpyHostToDevice, ctx->stream));
    AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  return AGZ_OK;
}

// the same Winograd-domain weights as two fp16 pieces (conv_wino_h2.hpp)
int agz_net::build_wino_h2_weights() {
  AGZ_REQUIRE(cfg == 0, AGZ_E_UNSUPPORTED, "agz_net: the Winograd path needs K a multiple of 64");
  wino_tm = agz::wino_h2_pick_tm(H, W);
  AGZ_REQUIRE((size_t)(wino_tm + 2) * (wino_tm + 2) * (Kp / 32) * 2 * (2 * Kp) * 64 < ((size_t)1 << 32), AGZ_E_UNSUPPORTED, "agz_net: K %d too wide for the Winograd weight image", conf.K);
  AGZ_HIP_TRY(hipSetDevice(ctx->device));
  for (auto& p : d_u2_dual) if (p) hipFree(p);
  for (auto& p : d_u2_tin) if (p) hipFree(p);
  for (auto& p : d_u2_colun) if (p) hipFree(p);
  d_u2_dual.assign(conf.SharedLayers, nullptr);
  d_u2_tin.assign(conf.SharedLayers, nullptr);
  d_u2_colun.assign(conf.SharedLayers, nullptr);
  free_u2c();
  const bool chained = agz::wino_h2c_ok(H, W, wino_tm, Kp);
  if (chained) { d_u2c_dual.assign(conf.SharedLayers, nullptr); wino_g1.assign(conf.SharedLayers, 0.f); wino_g0.assign(conf.SharedLayers, 0.f); }
  std::vector<_Float16> u2c;
  std::vector<std::vector<double>> l1s(conf.SharedLayers);   // per block [2 Kp]: sum_{ci,tap} |w[n][ci][tap]| / t_in[ci]
  u_unscale.assign(conf.SharedLayers, 1.0f);
  std::vector<float> tin, colun;
  std::vector<std::vector<float>> tins(conf.SharedLayers), coluns(conf.SharedLayers);
  for (auto& p : d_ep_h2) if (p) hipFree(p);
  d_ep_h2.assign(conf.SharedLayers, nullptr);
  if (d_ep_init_h2) { hipFree(d_ep_init_h2); d_ep_init_h2 = nullptr; }
  const int K = conf.K;
  size_t pi = 3;
  std::vector<_Float16> u2;
  for (int l = 0; l < conf.SharedLayers; l++, pi += 6) {
    const std::vector<float>& wa = params[pi].v;
    const std::vector<float>& wb = params[pi + 3].v;
    const int Kp_ = Kp;
    auto getw = [&](int n, int ci, int tap) -> double {
      const int o = n < Kp_ ? n : n - Kp_;
      if (o >= K || ci >= K) return 0.0;
      return (double)(n < Kp_ ? wa : wb)[((size_t)o * K + ci) * 9 + tap];
    };
    if (wino_tm == 5) agz::wino_build_u2<5>(u2, 2 * Kp, Kp, getw, tin, colun);
    else agz::wino_build_u2<4>(u2, 2 * Kp, Kp, getw, tin, colun);
    AGZ_HIP_TRY(hipMalloc(&d_u2_dual[l], u2.size() * 2));
    AGZ_HIP_TRY(hipMalloc(&d_u2_tin[l], tin.size() * 4));
    AGZ_HIP_TRY(hipMalloc(&d_u2_colun[l], colun.size() * 4));
    AGZ_HIP_TRY(hipMemcpyAsync(d_u2_tin[l], tin.data(), tin.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    AGZ_HIP_TRY(hipMemcpyAsync(d_u2_colun[l], colun.data(), colun.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    AGZ_HIP_TRY(hipMemcpyAsync(d_u2_dual[l], u2.data(), u2.size() * 2, hipMemcpyHostToDevice, ctx->stream));
    if (chained) {
      agz::wino_build_u2c(u2, u2c, (wino_tm + 2) * (wino_tm + 2), 2 * Kp, Kp, Kp);
      AGZ_HIP_TRY(hipMalloc(&d_u2c_dual[l], u2c.size() * 2));
      AGZ_HIP_TRY(hipMemcpyAsync(d_u2c_dual[l], u2c.data(), u2c.size() * 2, hipMemcpyHostToDevice, ctx->stream));
      l1s[l].assign(2 * Kp, 0.0);
      for (int n = 0; n < 2 * Kp; n++)
        for (int ci = 0; ci < Kp; ci++)
          for (int tap = 0; tap < 9; tap++) l1s[l][n] += std::fabs(getw(n, ci, tap)) / (double)tin[ci];
    }
    AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    tins[l] = tin; coluns[l] = colun;
  }
  // the tower's epilogue parameters with the scales folded in (net.hpp)
  AGZ_REQUIRE((int)h_ep_dual.size() == conf.SharedLayers && !h_ep_init.empty(), AGZ_E_STATE, "agz_net: Winograd weights before commit");
  const int hw = H * W;
  std::vector<float> e2;
  for (int l = 0; l < conf.SharedLayers; l++) {
    const std::vector<float>& e = h_ep_dual[l];
    const bool more = l + 1 < conf.SharedLayers;
    e2.assign(e.size(), 0.f);
    for (int p = 0; p < hw; p++)
      for (int c = 0; c < Kp; c++) {
        const float tn = more ? tins[l + 1][c] : 1.0f;
        const size_t o = ((size_t)p * Kp + c) * 4;
        e2[o + 0] = e[o + 0] * coluns[l][c] * tn; e2[o + 1] = e[o + 1] * tn;
        e2[o + 2] = e[o + 2] * coluns[l][Kp + c] * tn; e2[o + 3] = e[o + 3] * tn;
      }
    if (chained) {   // max |y t_next| <= g1 max|x t_in| + g0 over every pixel and channel (relu(s v + t) <= |s| |v| + max(t, 0))
      double g1 = 0.0, g0 = 0.0;
      for (int p = 0; p < hw; p++)
        for (int c = 0; c < Kp; c++) {
          const double tn = more ? tins[l + 1][c] : 1.0;
          const size_t o = ((size_t)p * Kp + c) * 4;
          g1 = std::max(g1, tn * (std::fabs((double)e[o + 0]) * l1s[l][c] + std::fabs((double)e[o + 2]) * l1s[l][Kp + c]));
          g0 = std::max(g0, tn * (std::max((double)e[o + 1], 0.0) + std::max((double)e[o + 3], 0.0)));
        }
      wino_g1[l] = (float)(g1 * (1.0 + 1e-6)); wino_g0[l] = (float)(g0 * (1.0 + 1e-6));
    }
    AGZ_HIP_TRY(hipMalloc(&d_ep_h2[l], e2.size() * 4));
    AGZ_HIP_TRY(hipMemcpyAsync(d_ep_h2[l], e2.data(), e2.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  if (conf.SharedLayers > 0) {
    e2.assign(h_ep_init.size(), 0.f);
    for (int p = 0; p < hw; p++)
      for (int c = 0; c < Kp; c++) {
        const size_t o = ((size_t)p * Kp + c) * 2;
        e2[o] = h_ep_init[o] * tins[0][c]; e2[o + 1] = h_ep_init[o + 1] * tins[0][c];
      }
    AGZ_HIP_TRY(hipMalloc(&d_ep_init_h2, e2.size() * 4));
    AGZ_HIP_TRY(hipMemcpyAsync(d_ep_init_h2, e2.data(), e2.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  return AGZ_OK;
}

agz_net::FwdPlan agz_net::fwd_plan(int B) const {
  const int M = B * HW;
  FwdPlan p{};
  // tile height: 128-row tiles are the most efficient; when a layer has fewer of them than half the CU slots
  // (Connect-4: 84 tiles on 512 slots) 64-row tiles spread the work over twice as many CUs (measured +5%)
  auto n_tiles128 = [&](int ntot) { return ceil_div(M, 128) * ceil_div(ntot, 128); };
  p.half_init = cfg == 0 && n_tiles128(Kp) < ctx->num_cus;
  p.half_dual = cfg == 0 && n_tiles128(2 * Kp) < ctx->num_cus;
  // latency regime (one decision per forward, so every layer and the heads agree): the dual layers would occupy
  // at most a quarter of the CUs.  Tournament Agent.Search (one tree, batch 1) lands here.
  const int tiles_dual = cfg != 0 ? ceil_div(M, 128) * ceil_div(2 * Kp, 64) : ceil_div(M, 64) * ceil_div(2 * Kp, 128);
  // ... or, for wide towers (K >= 256: 72 K-iterations per tile), would not even give every CU one tile: a round of 8-16
  // lanes of a single tree (agz_arena_set_parallel) — measured 1.135 -> 1.05 s per move at 8 lanes.  Narrow towers must
  // not take this branch (Connect-4, K=64, 168 tiles: 183 -> 139 games/s).
  p.small = latency_mode && Kp >= 64 &&
            (tiles_dual * 4 <= ctx->num_cus || (Kp >= 256 && tiles_dual <= ctx->num_cus));  // 32-wide towers are launch-bound
  // A split mode with AGZ_COMPUTE_FORCE keeps ITS tower at every batch size (so a lane round of 16 boards and the batch-1
  // prepareRoot of the same search run the same arithmetic, bit for bit per board): measured on G19T (40 blocks), per block:
  // 16 boards 0.086 ms Winograd fp16x2 vs 0.150 ms split-K fp32; 8 boards 0.068 vs 0.086; 1 board 0.053 vs 0.023 — the
  // batch-1 evaluation happens once per move, the rounds 100 times (profiles/r02/latency_modes.log).  The heads keep the
  // spread small-batch form either way.
  p.forced_split = compute_force && cfg == 0 && conf.SharedLayers > 0 &&
                   (this->compute_mode == AGZ_COMPUTE_WINO_H2 || this->compute_mode == AGZ_COMPUTE_WINO ||
                    this->compute_mode == AGZ_COMPUTE_BF16X3 || this->compute_mode == AGZ_COMPUTE_FP16X2);
  p.latency = p.small && !p.forced_split;          // split-K convolutions
  const int spread_max = 64;
  // (wide towers: the spread form also wins at 512 boards — 0.17 vs 0.35 ms at 19x19 / K=256, profiles/r02/init_heads_ab.log)
  p.heads_spread = p.small || (latency_mode && (B <= spread_max || Kp >= 256));
  p.heads_nb = B >= 32;   // heads_fc_nb_kernel<4> (four boards per workgroup) from 32 boards
  // the split kernels only pay once the 128-row tiles fill the chip (Connect-4, K=64, 256 games: 84 tiles -> the fp32
  // half-tile kernel is faster: 187 vs 162 / 158 games/s measured)
  // (the split kernels use 32-bit BYTE offsets: activation tensor below 4 GiB)
  p.split_ok = cfg == 0 && !p.latency && (!p.half_dual || compute_force) && conf.SharedLayers > 0 &&
               (size_t)B * Hp * Wp * Kp * sizeof(float) < ((size_t)1 << 32);
  // the three-kernel Winograd fp16x2 block picks its GEMM tile by how many workgroups the batch gives (forward_packed)
  {
    const int npos = (wino_tm + 2) * (wino_tm + 2), tpb = ceil_div(H, wino_tm) * ceil_div(W, wino_tm);
    p.wino_wide = (2 * Kp) % 256 == 0 && (size_t)npos * ceil_div(B * tpb, 128) * ((2 * Kp) / 256) >= (size_t)ctx->num_cus;
  }
  return p;
}

int agz_net::min_same_batch(int n, int G) const {
  if (n >= G) return G;
  const FwdPlan want = fwd_plan(G);
  auto same = [&](int B) {
    const FwdPlan q = fwd_plan(B);
    return q.half_init == want.half_init && q.half_dual == want.half_dual && q.small == want.small && q.forced_split == want.forced_split &&
           q.latency == want.latency && q.heads_spread == want.heads_spread && q.split_ok == want.split_ok && q.heads_nb == want.heads_nb && q.wino_wide == want.wino_wide;
  };
  if (same(n)) return n;
  for (int B = 16; B < G; B *= 2)
    if (B > n && same(B)) return B;
  return G;
}

int agz_net::forward_packed(int B, float* policy_dev, float* value_dev) {
  AGZ_REQUIRE(committed, AGZ_E_STATE, "agz_net: infer before agz_net_commit");
  AGZ_REQUIRE(B >= 1 && B <= max_batch, AGZ_E_INVALID, "agz_net: batch %d exceeds allocated %d", B, max_batch);
  ConvArgs a{};
  a.M = B * HW; a.HW = HW; a.W = W; a.Wp = Wp; a.HpWp = Hp * Wp;
  // K1: init conv  F -> K  (+BN+ReLU)
  a.x = d_act_in; a.w = d_w_init; a.ep = d_ep_init; a.y = d_actA;
  a.Cin_p = Fp; a.Cout_p = Kp; a.Ntot = Kp;
  const FwdPlan plan = fwd_plan(B);
  const bool half_init = plan.half_init, half_dual = plan.half_dual, latency = plan.latency, heads_spread = plan.heads_spread, split_ok = plan.split_ok;
  float** wsp = latency ? &d_ws : nullptr;
  // AGZ_COMPUTE_AUTO: the measured choice — the Winograd fp16x2 tower wherever its weights exist (K a multiple of 64; 19x19 K=256:
  // 0.72 vs 2.03 ms per block for bf16x3, 9x9 K=128: 21.4 vs 13.5 games/s), else bf16x3; shapes below the chip-filling threshold
  // keep the fp32 kernels either way
  int compute_mode = this->compute_mode;
  if (compute_mode == AGZ_COMPUTE_AUTO) compute_mode = !d_u2_dual.empty() ? AGZ_COMPUTE_WINO_H2 : AGZ_COMPUTE_BF16X3;
  const bool use_h2 = split_ok && compute_mode == AGZ_COMPUTE_FP16X2;
  const bool wino_ok = split_ok && compute_mode == AGZ_COMPUTE_WINO;
  const bool wino_h2_ok = split_ok && compute_mode == AGZ_COMPUTE_WINO_H2;
  if (wino_h2_ok && d_ep_init_h2) a.ep = d_ep_init_h2;   // the tower takes its input pre-scaled by block 0's t_in (net.hpp)
  if (use_h2 && (size_t)B > amax_cap) {
    AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (d_amax) hipFree(d_amax);
    d_amax = nullptr; amax_cap = 0;
    AGZ_HIP_TRY(hipMalloc(&d_amax, (size_t)B * sizeof(unsigned)));
    amax_cap = (size_t)B;
  }
  // latency regime with the one-launch-per-layer fp16x2 tower (conv_lat.hpp): the input layer, its epilogue and the first layer's range
  // words in one launch as well (lat_input_kernel)
  bool lat_h2 = latency && cfg == 0 && this->compute_mode != AGZ_COMPUTE_F32_MFMA && conf.SharedLayers > 0 &&
                (int)d_lat_w2.size() == conf.SharedLayers && agz::conv_lat_ok(Kp, Kp, Wp) && ceil_div(HW, agz::LAT_ROWS) * (Kp / 8) <= 256;
  for (int l = 0; lat_h2 && l < conf.SharedLayers; l++) lat_h2 = d_lat_w2[l] != nullptr;
  const int in_groups = ceil_div(HW, agz::LAT_IN_PIX), in_words = in_groups * (Kp / 64);
  const bool lat_in = lat_h2 && Fp == 32 && Kp % 64 == 0 && in_words <= 256 && d_w_init_t;
  if (lat_h2 && lat_wmax_cap < B) {
    AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
    for (auto& p2 : d_lat_wmax) { if (p2) hipFree(p2); p2 = nullptr; }
    lat_wmax_cap = 0;
    for (auto& p2 : d_lat_wmax) AGZ_HIP_TRY(hipMalloc(&p2, (size_t)B * 256 * sizeof(float)));
    lat_wmax_cap = B;
  }
  int rc;
  bool amax0_done = false;   // the input layer's epilogue produced the tower's per-board input ranges (d_amax[0 .. B))
  if (lat_in) {
    agz::LatInArgs li{};
    li.x = d_act_in; li.w = d_w_init_t; li.ep = reinterpret_cast<const float2*>(d_ep_init); li.t = d_lat_tin[0]; li.y = d_actA; li.words = d_lat_wmax[1];
    li.B = B; li.H = H; li.W = W; li.Hp = Hp; li.Wp = Wp; li.Cout_p = Kp; li.groups_per_board = in_groups;
    ProfScope ps(ctx, AGZ_PROF_CONV_INIT);
    hipLaunchKernelGGL(agz::lat_input_kernel, dim3(Kp / 64, B * in_groups), dim3(256), 0, ctx->stream, li);
    rc = AGZ_OK;
  }
  else if (split_ok && compute_mode != AGZ_COMPUTE_F32_MFMA && d_w3_init && Kp % 128 == 0 &&
      (size_t)B * Hp * Wp * Fp * sizeof(float) < ((size_t)1 << 32)) {
    // the split modes: input convolution with bf16x3 products too (same fp32-grade arithmetic as AGZ_COMPUTE_BF16X3)
    a.n_ntiles = ceil_div(a.Ntot, 128); a.n_mtiles = ceil_div(a.M, 128);
    a.splits = 1; a.per = 0; a.ws = nullptr; a.raw = 0;
    ProfScope ps(ctx, AGZ_PROF_CONV_INIT);
    // Winograd fp16x2 tower: the per-board range of its input comes out of this epilogue (one atomic per half wave and board) instead
    // of a sweep over the stored tensor (board_amax_kernel: 56 us per 512-board pass); the words must exist already (any later forward)
    if (wino_h2_ok && d_amax && amax_cap >= (size_t)B) {
      AGZ_HIP_TRY(hipMemsetAsync(d_amax, 0, (size_t)B * sizeof(unsigned), ctx->stream));
      a.amax_out = d_amax; amax0_done = true;
    }
    hipLaunchKernelGGL((conv3x3_x3_kernel<false>), dim3(a.n_mtiles * a.n_ntiles), dim3(256), 0, ctx->stream, a, d_w3_init);
    a.amax_out = nullptr;
    rc = AGZ_OK;
  }
  else if (cfg != 0) rc = launch_conv<4, 1, 1, false>(ctx, a, wsp, &ws_cap);
  else if (half_init) rc = launch_conv<2, 2, 1, false>(ctx, a, wsp, &ws_cap);
  else rc = launch_conv<2, 2, 2, false>(ctx, a, wsp, &ws_cap);
  if (rc != AGZ_OK) return rc;
  // K2: SharedLayers x fused dual-branch block
  float* cur = d_actA;
  float* nxt = d_actB;
  bool tower_done = false;
  if (!tower_done && wino_h2_ok) {
    // Winograd with fp16x2 transform-domain products (conv_wino_h2.hpp): per-board ranges in d_amax[block][board]
    AGZ_REQUIRE((int)d_u2_dual.size() == conf.SharedLayers, AGZ_E_STATE, "agz_net: Winograd fp16x2 weights not built");
    // GEMM form (measured on G19/B=512, profiles/r02/wino_h2_gemm_variants.log): 128x256 tile with the A operand fetched two
    // steps ahead 0.385 ms, 128x128 0.40 ms, the plain single-prefetch kernels 0.47-0.50 ms.
    const int npos = (wino_tm + 2) * (wino_tm + 2);
    const int tpb = ceil_div(H, wino_tm) * ceil_div(W, wino_tm);
    // ... and the 128-column tile when the 256-column grid would leave CUs without a workgroup (a lane round of 16 boards: 196
    // against 392 workgroups, 0.0747 -> 0.0726 ms per block, p50 move 0.250 -> 0.241 s)
    const bool wide = plan.wino_wide;
    // Board chunks and queues (agz_net_set_tower_queues; AGZ_WINO_H2_CHUNK = boards per chunk, AGZ_WINO_H2_QUEUES = 1 | 2 override):
    // chunk i runs its block chain on queue i % queues with that queue's scratch — chains of different boards are independent
    // (per-board ranges, bit-identical results), so one half-batch's HBM-bound transform kernels run under the other's GEMM and
    // fill the GEMM's last partial round of workgroups: 15.39 -> 14.51 ms per 512-board pass (round 2), the default from 256 boards.
    static const int chunk_env = [] { const char* e = getenv("AGZ_WINO_H2_CHUNK"); return e ? atoi(e) : 0; }();
    static const int queues_env = [] { const char* e = getenv("AGZ_WINO_H2_QUEUES"); return e ? atoi(e) : 0; }();
    const int queues_want = queues_env > 0 ? queues_env : (tower_queues > 0 ? tower_queues : (B >= 256 ? 2 : 1));
    // 32-bit byte offsets into V: npos * (tiles rounded up to 128 + pad) * Kp * 4 < 2^32
    const int chunk_max = (int)std::min<size_t>((size_t)B, ((((size_t)1 << 32) - 1) / ((size_t)npos * Kp * 4) - 127) / tpb);
    int chunk = chunk_env >= 1 ? std::min(chunk_env, chunk_max) : chunk_max;
    const int ns = (queues_want == 2 && B >= 64) ? 2 : 1;
    if (ns == 2 && chunk >= B) chunk = (B + 1) / 2;
    const size_t v_elems = wino_h2_rows(npos, (size_t)chunk * tpb) * Kp, m_elems = 2 * v_elems;
    // Chained form (conv_wino_h2c.hpp; AGZ_WINO_H2_FORM = 0 keeps the three-kernel block): output transform of block l and input
    // transform of block l+1 in one kernel, y never leaves the chip between blocks.  V2(l+1) of a chunk lives in scratch from one
    // block to the next, so every CHUNK keeps its own scratch there (the three-kernel block: every queue)
    static const int form_env = [] { const char* e = getenv("AGZ_WINO_H2_FORM"); return e ? atoi(e) : -1; }();
    const int form_want = wino_form >= 0 ? wino_form : form_env;
    const bool chained = form_want != 0 && (int)d_u2c_dual.size() == conf.SharedLayers && agz::wino_h2c_ok(H, W, wino_tm, Kp);
    const int n_scr = chained ? ceil_div(B, chunk) : ns;
    if (v_elems * n_scr > wino_v_cap) {
      AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
      if (ctx->stream2) AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream2));
      if (d_wV) hipFree(d_wV);
      if (d_wM) hipFree(d_wM);
      d_wV = d_wM = nullptr; wino_chunk_cap = 0;
      AGZ_HIP_TRY(hipMalloc(&d_wV, v_elems * n_scr * sizeof(float)));
      AGZ_HIP_TRY(hipMalloc(&d_wM, m_elems * n_scr * sizeof(float)));
      wino_v_cap = v_elems * n_scr; wino_chunk_cap = 0;   // (the fp32-V Winograd path sizes by boards: force its re-allocation)
    }
    // per-board ranges [blocks+1][B], then the per-wave maxima of the output kernel [queues][chunk tiles][Kp/64] (as floats)
    // per-wave maxima of the output kernel: every chunk of boards keeps its own region from one block to the next (the next
    // block's input transform reduces them), one word per tile and 64 channels
    const size_t wm_board = chained ? (size_t)tpb * (Kp >> 5) * 2 : (size_t)tpb * (Kp >> 6);   // (chained: two arrays, ping-pong)
    const size_t need_amax = (size_t)(conf.SharedLayers + 1) * B + wm_board * B;
    if (need_amax > amax_cap) {
      AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
      if (ctx->stream2) AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream2));
      if (d_amax) hipFree(d_amax);
      d_amax = nullptr; amax_cap = 0;
      AGZ_HIP_TRY(hipMalloc(&d_amax, need_amax * sizeof(unsigned)));
      amax_cap = need_amax;
      amax0_done = false;   // (the words just written went with the old buffer)
    }
    float* d_wave_max = reinterpret_cast<float*>(d_amax + (size_t)(conf.SharedLayers + 1) * B);
    if (!amax0_done)
      hipLaunchKernelGGL(board_amax_kernel, dim3(B), dim3(256), 0, ctx->stream, cur, d_amax, HW, W, Wp, Hp * Wp, Kp, (const float*)nullptr);   // (pre-scaled input)
    if (ns == 2) {
      if (!ctx->stream2) {
        AGZ_HIP_TRY(hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking));
        AGZ_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
        AGZ_HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
      }
      AGZ_HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
      AGZ_HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->ev_fork, 0));
    }
    for (int l = 0; chained && l < conf.SharedLayers; l++) {
      ProfScope ps(ctx, AGZ_PROF_CONV);
      int ci = 0;
      const size_t wmb = (size_t)tpb * (Kp >> 5);
      const bool last = l + 1 == conf.SharedLayers;
      for (int b0 = 0; b0 < B; b0 += chunk, ci++) {
        const int q = ci % ns;
        hipStream_t st = q ? ctx->stream2 : ctx->stream;
        WinoH2Args hh{};
        WinoArgs& wa = hh.w;
        wa.x = cur + (size_t)b0 * Hp * Wp * Kp; wa.y = nxt + (size_t)b0 * Hp * Wp * Kp;   // x: block 0 only; y: the last block only
        wa.V = d_wV + (size_t)ci * v_elems; wa.Mb = d_wM + (size_t)ci * m_elems; wa.ep = d_ep_h2[l];   // (the chunk's own scratch)
        wa.B = std::min(chunk, B - b0); wa.H = H; wa.W = W; wa.Hp = Hp; wa.Wp = Wp; wa.C = Kp; wa.Cout_p = Kp; wa.Ntot = 2 * Kp;
        hh.U2c = d_u2c_dual[l]; hh.w_unscale = 1.f; hh.tm = wino_tm;
        hh.amax_in = d_amax + (size_t)l * B + b0;                                       // the range word V2(l) was written with
        hh.amax_next = last ? nullptr : d_amax + (size_t)(l + 1) * B + b0;
        hh.amax_true = l == 0 ? reinterpret_cast<const float*>(d_amax + b0) : nullptr;  // block 0: board_amax_kernel's exact word
        hh.wm_prev = d_wave_max + ((size_t)((l + 1) & 1) * B + b0) * wmb;
        hh.wm_out = d_wave_max + ((size_t)(l & 1) * B + b0) * wmb;
        hh.g1 = wino_g1[l]; hh.g0 = wino_g0[l];
        {   // which GEMM kernel / store policy: agz_net_set_wino_h2_gemm, else AGZ_WINO_H2_GEMM (1 | 2, + 64 = round 4's stores), else the default
          // (the environment reaches the two product kernels only: 1 | 2, + 64; anything else — 2 + 16 * mode are the persistent kernel's
          //  timing-only decomposition instances, wrong results by design, agz_debug.h — is ignored with one line on stderr)
          static const int gemm_env = [] {
            const char* e = getenv("AGZ_WINO_H2_GEMM");
            const int v = e ? atoi(e) : 0;
            if (v == 0 || (((v & 63) == 1 || (v & 63) == 2) && (v >> 6) <= 1)) return v;
            fprintf(stderr, "libagz: AGZ_WINO_H2_GEMM=%s ignored (want 1 or 2, optionally + 64)\n", e);
            return 0;
          }();
          const int gv = wino_gemm > 0 ? wino_gemm : gemm_env;
          hh.gemm_variant = gv & 63; hh.temporal_stores = (gv >> 6) & 1;
        }
        if (l == 0) agz::wino_h2c_in(ctx, hh, st);
        agz::wino_h2c_gemm(ctx, hh, st);
        agz::wino_h2c_oi(ctx, hh, last, st, form_want == 1 ? 1 : 4);   // (A/B hook: form 1 = the plain out->in kernel)
      }
      if (last) std::swap(cur, nxt);
    }
    for (int l = 0; !chained && l < conf.SharedLayers; l++) {
      ProfScope ps(ctx, AGZ_PROF_CONV);
      int ci = 0;
      for (int b0 = 0; b0 < B; b0 += chunk, ci++) {
        const int q = ci % ns;
        WinoH2Args hh{};
        WinoArgs& wa = hh.w;
        wa.x = cur + (size_t)b0 * Hp * Wp * Kp; wa.y = nxt + (size_t)b0 * Hp * Wp * Kp;
        wa.V = d_wV + (size_t)q * v_elems; wa.Mb = d_wM + (size_t)q * m_elems; wa.ep = d_ep_h2[l];
        wa.B = std::min(chunk, B - b0); wa.H = H; wa.W = W; wa.Hp = Hp; wa.Wp = Wp; wa.C = Kp; wa.Cout_p = Kp; wa.Ntot = 2 * Kp;
        hh.U2 = d_u2_dual[l]; hh.w_unscale = 1.f; hh.tm = wino_tm;
        hh.t_in = nullptr; hh.col_unscale = nullptr; hh.t_next = nullptr;   // all folded into d_ep_h2 / d_ep_init_h2: activations travel pre-scaled
        hh.amax_in = d_amax + (size_t)l * B + b0; hh.amax_out = d_amax + (size_t)(l + 1) * B + b0;
        hh.wave_max = d_wave_max + (size_t)b0 * wm_board;
        hh.fuse_prev = l > 0;   // block 0's input range comes from board_amax_kernel above
        wino_h2_launch(ctx, hh, wide, q ? ctx->stream2 : ctx->stream);
      }
      std::swap(cur, nxt);
    }
    if (ns == 2) {
      AGZ_HIP_TRY(hipEventRecord(ctx->ev_join, ctx->stream2));
      AGZ_HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
    }
    tower_done = true;
  }
  for (int l = 0; !tower_done && l < conf.SharedLayers; l++) {
    a.x = cur; a.w = d_w_dual[l]; a.ep = d_ep_dual[l]; a.y = nxt;
    a.Cin_p = Kp; a.Cout_p = Kp; a.Ntot = 2 * Kp;
    if (use_h2) {
      a.n_ntiles = ceil_div(a.Ntot, 128); a.n_mtiles = ceil_div(a.M, 128);
      a.splits = 1; a.per = 0; a.ws = nullptr; a.raw = 0;
      // per-board range of this layer's input (one word per board: results stay independent of the batch composition)
      hipLaunchKernelGGL(board_amax_kernel, dim3(B), dim3(256), 0, ctx->stream, cur, d_amax, HW, W, Wp, Hp * Wp, Kp);
      a.amax_in = d_amax; a.w_unscale = w_unscale[l];
      ProfScope ps(ctx, AGZ_PROF_CONV);
      if ((2 * Kp) % 256 == 0) {   // wide tile: 128 x 256
        a.n_ntiles = (2 * Kp) / 256;
        hipLaunchKernelGGL(conv3x3_h2w_kernel, dim3(a.n_mtiles * a.n_ntiles), dim3(256), 0, ctx->stream, a, d_w2_dual[l]);
      } else {
        hipLaunchKernelGGL(conv3x3_h2_kernel, dim3(a.n_mtiles * a.n_ntiles), dim3(256), 0, ctx->stream, a, d_w2_dual[l]);
      }
      rc = AGZ_OK;
    }
    else if (wino_ok) {
      // Winograd F(4x4,3x3): boards in chunks (scratch V + M: 2.8 MB per 19x19 board at K=256)
      AGZ_REQUIRE((int)d_u3_dual.size() == conf.SharedLayers && d_u3_dual[l], AGZ_E_STATE, "agz_net: Winograd weights not built");
      static const int chunk_env = [] { const char* e = getenv("AGZ_WINO_CHUNK"); return e ? atoi(e) : 0; }();  // tuning knob
      const int tpb = ceil_div(H, 4) * ceil_div(W, 4);
      // 32-bit byte offsets into V inside the GEMM: 36 * T * Kp * 4 < 4 GiB
      const int chunk_max = (int)std::min<size_t>((size_t)B, (((size_t)1 << 32) - 1) / ((size_t)36 * tpb * Kp * 4));
      const int chunk = chunk_env >= 1 ? std::min(chunk_env, chunk_max) : chunk_max;
      if (chunk > wino_chunk_cap) {
        AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (d_wV) hipFree(d_wV);
        if (d_wM) hipFree(d_wM);
        d_wV = d_wM = nullptr; wino_chunk_cap = 0;
        AGZ_HIP_TRY(hipMalloc(&d_wV, (size_t)36 * chunk * tpb * Kp * sizeof(float)));
        AGZ_HIP_TRY(hipMalloc(&d_wM, (size_t)36 * chunk * tpb * 2 * Kp * sizeof(float)));
        wino_chunk_cap = chunk; wino_v_cap = 0;
      }
      ProfScope ps(ctx, AGZ_PROF_CONV);
      for (int b0 = 0; b0 < B; b0 += chunk) {
        WinoArgs wa{};
        wa.x = cur + (size_t)b0 * Hp * Wp * Kp; wa.y = nxt + (size_t)b0 * Hp * Wp * Kp;
        wa.V = d_wV; wa.Mb = d_wM; wa.U3 = d_u3_dual[l]; wa.ep = d_ep_dual[l];
        wa.B = std::min(chunk, B - b0); wa.H = H; wa.W = W; wa.Hp = Hp; wa.Wp = Wp; wa.C = Kp; wa.Cout_p = Kp; wa.Ntot = 2 * Kp;
        wino_launch(ctx, wa, WINO_IN | WINO_GEMM | WINO_OUT);
      }
      rc = AGZ_OK;
    }
    else if (split_ok && compute_mode == AGZ_COMPUTE_BF16X3) {
      a.n_ntiles = ceil_div(a.Ntot, 128); a.n_mtiles = ceil_div(a.M, 128);
      a.splits = 1; a.per = 0; a.ws = nullptr; a.raw = 0;
      ProfScope ps(ctx, AGZ_PROF_CONV);
      hipLaunchKernelGGL((conv3x3_x3_kernel<true>), dim3(a.n_mtiles * a.n_ntiles), dim3(256), 0, ctx->stream, a, d_w3_dual[l]);
      rc = AGZ_OK;
    }
    // (fp16x2 products: only in the modes whose contract is split operands — AGZ_COMPUTE_F32_MFMA keeps exact fp32 products at
    //  every batch size and takes the split-K fp32 kernels below)
    else if (lat_h2) {   // (range words per board: four per lane)
      // latency regime, one launch per layer: K split inside the workgroup, weights up front, no partial sums in memory; fp16x2
      // products on equilibrated operands (conv_lat.hpp).  Range words: layer 0 from a board reduction, then from layer to layer.
      const int gpb = ceil_div(HW, agz::LAT_ROWS), words = gpb * (Kp / 8);
      agz::LatH2Args la{};
      la.x = cur; la.w2 = d_lat_w2[l]; la.t_in = d_lat_tin[l]; la.col_unscale = d_lat_colun[l]; la.ep = d_ep_dual[l]; la.y = nxt;
      la.B = B; la.H = H; la.W = W; la.Hp = Hp; la.Wp = Wp; la.C = Kp; la.Cout_p = Kp; la.Ntot = 2 * Kp;
      la.groups_per_board = gpb;
      if (l == 0) {
        if (!lat_in) hipLaunchKernelGGL(agz::lat_board_words_kernel, dim3(B * 64), dim3(256), 0, ctx->stream, cur, (const float*)d_lat_tin[0], d_lat_wmax[1], HW, W, Wp, Hp * Wp, Kp, 64);
        la.wmax_in = d_lat_wmax[1]; la.n_in_words = lat_in ? in_words : 64;
      } else {
        la.wmax_in = d_lat_wmax[(l - 1) & 1]; la.n_in_words = words;
      }
      const bool more = l + 1 < conf.SharedLayers;
      la.t_next = more ? d_lat_tin[l + 1] : nullptr;
      la.wmax_out = more ? d_lat_wmax[l & 1] : nullptr;
      ProfScope ps(ctx, AGZ_PROF_CONV);
      agz::conv_lat_h2_launch(ctx, la);
      rc = AGZ_OK;
    }
    else if (cfg != 0) rc = launch_conv<4, 1, 1, true>(ctx, a, wsp, &ws_cap);
    else if (half_dual) rc = launch_conv<2, 2, 1, true>(ctx, a, wsp, &ws_cap);
    else rc = launch_conv<2, 2, 2, true>(ctx, a, wsp, &ws_cap);
    if (rc != AGZ_OK) return rc;
    std::swap(cur, nxt);
  }
  // K4+K5 heads
  HeadArgs h{};
  h.x = cur; h.conv = d_head_conv; h.bn = d_head_bn; h.Wp = d_Wp; h.bp = d_bp; h.W1 = d_W1; h.b1 = d_b1; h.W2 = d_W2;
  h.b2 = d_b2; h.policy = policy_dev; h.value = value_dev;
  h.H = H; h.W = W; h.HW = HW; h.Wp_ = Wp; h.HpWp = Hp * Wp; h.Kp = Kp; h.A = conf.ActionSpace; h.FC = conf.FC;
  size_t smem = (size_t)(3 * HW + 8 + conf.FC) * sizeof(float);
  {
    ProfScope ps(ctx, AGZ_PROF_HEADS);
    if (heads_spread) {
      // few boards: one workgroup per board would leave the 1.8 MB of FC weights to a single CU (0.26 ms at 19x19);
      // spread the 1x1 convs over pixels and the two FC layers over output columns instead
      size_t need = (size_t)B * (3 * HW + conf.ActionSpace + conf.FC);
      if (need > hs_cap) {
        AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (d_hs) hipFree(d_hs);
        d_hs = nullptr; hs_cap = 0;
        AGZ_HIP_TRY(hipMalloc(&d_hs, need * sizeof(float)));
        hs_cap = need;
      }
      h.feat = d_hs; h.cols = d_hs + (size_t)B * 3 * HW;
      hipLaunchKernelGGL(heads_feat_kernel, dim3(ceil_div(HW, 4), B), dim3(256), 0, ctx->stream, h);
      if (plan.heads_nb)
        hipLaunchKernelGGL(heads_fc_nb_kernel<4>, dim3(ceil_div(conf.ActionSpace + conf.FC, 64), ceil_div(B, 4)), dim3(1024), 0, ctx->stream, h, B);
      else
        hipLaunchKernelGGL(heads_fc_kernel, dim3(ceil_div(conf.ActionSpace + conf.FC, 64), B), dim3(1024), 0, ctx->stream, h);
      hipLaunchKernelGGL(heads_out_kernel, dim3(B), dim3(256), 0, ctx->stream, h);
    } else {
      hipLaunchKernelGGL(heads_kernel, dim3(B), dim3(256), smem, ctx->stream, h);
    }
  }
  AGZ_HIP_TRY(hipGetLastError());
  return AGZ_OK;
}

int agz_net::forward_dev(const float* planes_dev, int B, float* policy_dev, float* value_dev) {
  AGZ_REQUIRE(committed, AGZ_E_STATE, "agz_net: infer before agz_net_commit");
  int r = ensure_batch(B);
  if (r != AGZ_OK) return r;
  int total = B * HW * Fp;
  hipLaunchKernelGGL(pack_planes_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, ctx->stream, planes_dev, d_act_in, B,
                     conf.Features, H, W, Fp);
  return forward_packed(B, policy_dev, value_dev);
}

// ------------------------------------------------------------------------------------------------
extern "C" {

int agz_net_create(agz_ctx* ctx, const agz_net_conf* c, agz_net** out) {
  AGZ_REQUIRE(ctx && c && out, AGZ_E_INVALID, "agz_net_create: NULL argument");
  // dual.Config.IsValid (dualnet/config.go:33-42)
  AGZ_REQUIRE(c->K >= 1 && c->ActionSpace >= 3 && c->SharedLayers >= 0 && c->FC > 1 && c->BatchSize >= 1 && c->Features > 0,
              AGZ_E_INVALID, "agz_net_create: NNConf is not valid");
  AGZ_REQUIRE(c->Width >= 1 && c->Height >= 1, AGZ_E_INVALID, "agz_net_create: bad board size");
  AGZ_REQUIRE(c->Features <= 32, AGZ_E_UNSUPPORTED, "agz_net_create: Features > 32 unsupported");
  AGZ_REQUIRE(c->ActionSpace <= 512, AGZ_E_UNSUPPORTED, "agz_net_create: ActionSpace > 512 unsupported");
  AGZ_REQUIRE(c->bn_mode >= 0 && c->bn_mode <= 2, AGZ_E_INVALID, "agz_net_create: bad bn_mode");
  agz_net* n = new agz_net();
  n->ctx = ctx;
  n->conf = *c;
  n->H = c->Height; n->W = c->Width; n->HW = n->H * n->W; n->Hp = n->H + 2; n->Wp = n->W + 2;
  n->Kp = round_up(c->K, 32);
  n->cfg = (n->Kp % 64 == 0) ? 0 : 1;
  int K = c->K, F = c->Features, H = n->H, W = n->W, hw = n->HW, B = c->BatchSize;
  auto conv = [&](const std::string& nm, int o, int i, int k) {
    n->params.push_back(Param{"Filter" + nm, std::vector<float>((size_t)o * i * k * k, 0.f), {o, i, k, k}, 0});
  };
  auto bnp = [&](const std::string& nm, int C) {
    n->params.push_back(Param{nm + "_gamma", std::vector<float>((size_t)C * hw, 0.f), {B, C, H, W}, 1});
    n->params.push_back(Param{nm + "_beta", std::vector<float>((size_t)C * hw, 0.f), {B, C, H, W}, 1});
    n->bn.push_back(BNStats{std::vector<float>(C, 0.f), std::vector<float>(C, 1.f)});
  };
  auto fc = [&](const std::string& nm, int in, int units) {
    n->params.push_back(Param{nm + "_w", std::vector<float>((size_t)in * units, 0.f), {in, units}, 2});
    n->params.push_back(Param{nm + "_b", std::vector<float>((size_t)units, 0.f), {B, units}, 3});
  };
  conv("Init", K, F, 3); bnp("Init", K);
  for (int i = 0; i < c->SharedLayers; i++) {
    std::string s = std::to_string(i);
    conv("Layer1 of Shared Layer " + s, K, K, 3); bnp("L1_" + s, K);
    conv("Layer2 of Shared Layer " + s, K, K, 3); bnp("L2_" + s, K);
  }
  conv("PolicyHead", 2, K, 1); bnp("PolicyHead", 2);
  fc("Policy", 2 * hw, c->ActionSpace);
  conv("ValueHead", 1, K, 1); bnp("ValueHead", 1);
  fc("Value", hw, c->FC);
  fc("ValueOutput", c->FC, 1);
  *out = n;
  return AGZ_OK;
}

void agz_net_destroy(agz_net* n) {
  if (!n) return;
  hipStreamSynchronize(n->ctx->stream);
  n->free_device();
  delete n;
}

int agz_net_num_params(const agz_net* n) { return n ? (int)n->params.size() : 0; }

int agz_net_param_info(const agz_net* n, int i, char* name, size_t cap, size_t* n_elems) {
  AGZ_REQUIRE(n && i >= 0 && i < (int)n->params.size(), AGZ_E_INVALID, "agz_net_param_info: bad index %d", i);
  if (name && cap) { strncpy(name, n->params[i].name.c_str(), cap - 1); name[cap - 1] = 0; }
  if (n_elems) *n_elems = n->params[i].v.size();
  return AGZ_OK;
}

int agz_net_set_param(agz_net* n, int i, const float* host, size_t cnt) {
  AGZ_REQUIRE(n && host && i >= 0 && i < (int)n->params.size(), AGZ_E_INVALID, "agz_net_set_param: bad argument");
  Param& p = n->params[i];
  if (p.kind == 1 && cnt == (size_t)p.ref_shape[1]) {  // per-channel broadcast
    int C = p.ref_shape[1], hw = n->HW;
    for (int c = 0; c < C; c++) for (int q = 0; q < hw; q++) p.v[(size_t)c * hw + q] = host[c];
  } else {
    AGZ_REQUIRE(cnt >= p.v.size(), AGZ_E_INVALID, "agz_net_set_param(%s): need >= %zu floats, got %zu", p.name.c_str(),
                p.v.size(), cnt);
    memcpy(p.v.data(), host, p.v.size() * sizeof(float));  // row 0 of a batch-shaped tensor
  }
  n->committed = false;
  return AGZ_OK;
}

int agz_net_get_param(const agz_net* n, int i, float* host, size_t cnt) {
  AGZ_REQUIRE(n && host && i >= 0 && i < (int)n->params.size(), AGZ_E_INVALID, "agz_net_get_param: bad argument");
  const Param& p = n->params[i];
  AGZ_REQUIRE(cnt >= p.v.size(), AGZ_E_INVALID, "agz_net_get_param: buffer too small");
  memcpy(host, p.v.data(), p.v.size() * sizeof(float));
  return AGZ_OK;
}

int agz_net_set_bn_stats(agz_net* n, int bi, const float* mean, const float* var, size_t C) {
  AGZ_REQUIRE(n && mean && var && bi >= 0 && bi < (int)n->bn.size(), AGZ_E_INVALID, "agz_net_set_bn_stats: bad argument");
  AGZ_REQUIRE(C == n->bn[bi].mean.size(), AGZ_E_INVALID, "agz_net_set_bn_stats: C mismatch");
  n->bn[bi].mean.assign(mean, mean + C);
  n->bn[bi].var.assign(var, var + C);
  n->committed = false;
  return AGZ_OK;
}

int agz_net_init_random(agz_net* n, uint64_t seed) {
  AGZ_REQUIRE(n, AGZ_E_INVALID, "net is NULL");
  SplitMix64 r(seed);
  for (Param& p : n->params) {
    double field = 1;
    for (size_t i = 2; i < p.ref_shape.size(); i++) field *= p.ref_shape[i];
    double fan = (double)(p.ref_shape[0] + p.ref_shape[1]) * field;
    double stdev = std::sqrt(2.0 / fan);
    if (p.kind == 0) {  // GlorotU(1.0), ermahagerdmonards.go:39
      double lim = stdev * std::sqrt(3.0);
      for (float& x : p.v) x = (float)((r.float64() * 2.0 - 1.0) * lim);
    } else if (p.kind == 1 || p.kind == 2) {  // GlorotN(1.0), ermahagerdmonards.go:80 (+ BN gamma/beta, App. B b3)
      for (size_t i = 0; i < p.v.size(); i += 2) {
        double u1 = 1.0 - r.float64(), u2 = r.float64();
        double rad = std::sqrt(-2.0 * std::log(u1)), th = 6.283185307179586476925 * u2;
        p.v[i] = (float)(rad * std::cos(th) * stdev);
        if (i + 1 < p.v.size()) p.v[i + 1] = (float)(rad * std::sin(th) * stdev);
      }
    } else {  // Zeroes, ermahagerdmonards.go:82
      for (float& x : p.v) x = 0.f;
    }
  }
  n->committed = false;
  return AGZ_OK;
}

static int upload(float** dptr, const std::vector<float>& h, hipStream_t s) {
  if (*dptr) { hipFree(*dptr); *dptr = nullptr; }
  AGZ_HIP_TRY(hipMalloc(dptr, h.size() * sizeof(float)));
  AGZ_HIP_TRY(hipMemcpyAsync(*dptr, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipStreamSynchronize(s));  // h may be a temporary
  return AGZ_OK;
}

int agz_net_commit(agz_net* n) {
  AGZ_REQUIRE(n, AGZ_E_INVALID, "net is NULL");
  const agz_net_conf& c = n->conf;
  const int K = c.K, F = c.Features, hw = n->HW, Kp = n->Kp, Fp = n->Fp, A = c.ActionSpace, FCn = c.FC;
  hipStream_t s = n->ctx->stream;
  AGZ_HIP_TRY(hipSetDevice(n->ctx->device));
  // BN fold: y = ((x-mean)*inv)*gamma + beta  ->  x*scale + shift
  auto fold = [&](int bi, int ch, const Param& g, const Param& b, int p, float* scale, float* shift) {
    float mean = 0.f, inv = 1.f;
    if (c.bn_mode == AGZ_BN_DEGENERATE_EPS) inv = 1.0f / std::sqrt(0.0f + c.bn_eps);
    else if (c.bn_mode == AGZ_BN_RUNNING) { mean = n->bn[bi].mean[ch]; inv = 1.0f / std::sqrt(n->bn[bi].var[ch] + c.bn_eps); }
    float gm = g.v[(size_t)ch * hw + p], bt = b.v[(size_t)ch * hw + p];
    *scale = inv * gm;
    *shift = bt - mean * inv * gm;
  };
  size_t pi = 0;
  int bi = 0;
  // --- init conv: Wt[tap][n<Kp][c<Fp]
  {
    const Param& w = n->params[pi];
    std::vector<float> wt((size_t)9 * Kp * Fp, 0.f);
    for (int o = 0; o < K; o++) for (int ci = 0; ci < F; ci++) for (int t = 0; t < 9; t++)
      wt[((size_t)t * Kp + o) * Fp + ci] = w.v[((size_t)o * F + ci) * 9 + t];
    std::vector<float> ep((size_t)hw * Kp * 2, 0.f);
    for (int p = 0; p < hw; p++) for (int o = 0; o < K; o++)
      fold(bi, o, n->params[pi + 1], n->params[pi + 2], p, &ep[((size_t)p * Kp + o) * 2], &ep[((size_t)p * Kp + o) * 2 + 1]);
    int r;
    if ((r = upload(&n->d_w_init, wt, s)) != AGZ_OK) return r;
    {
      std::vector<float> wtt((size_t)9 * Fp * Kp, 0.f);
      for (int t = 0; t < 9; t++) for (int o = 0; o < Kp; o++) for (int ci = 0; ci < Fp; ci++)
        wtt[((size_t)t * Fp + ci) * Kp + o] = wt[((size_t)t * Kp + o) * Fp + ci];
      if ((r = upload(&n->d_w_init_t, wtt, s)) != AGZ_OK) return r;
    }
    if ((r = upload(&n->d_ep_init, ep, s)) != AGZ_OK) return r;
    n->h_ep_init = ep;
    if (n->cfg == 0) {
      // bf16x3 image of the input filter for conv3x3_x3_kernel<false>: w3[cc16][tap][piece][n][16] (exact truncation split);
      // in the split modes the F -> K input convolution runs on the bf16 pipe too (0.33 -> 0.1x ms at B=512: it pads F=18 to 32
      // channels on the slow fp32 MFMA otherwise)
      const int NC16 = Fp / 16;
      std::vector<unsigned short> w3((size_t)9 * NC16 * 3 * Kp * 16, 0);
      for (int t = 0; t < 9; t++) for (int nn = 0; nn < Kp; nn++) for (int ci = 0; ci < Fp; ci++) {
        float v = wt[((size_t)t * Kp + nn) * Fp + ci];
        uint32_t u, hu, mu, lu; memcpy(&u, &v, 4);
        hu = u & 0xffff0000u; float hf; memcpy(&hf, &hu, 4);
        float r1 = v - hf; uint32_t ru; memcpy(&ru, &r1, 4);
        mu = ru & 0xffff0000u; float mf; memcpy(&mf, &mu, 4);
        float r2 = r1 - mf; memcpy(&lu, &r2, 4);
        size_t base = (((size_t)((ci / 16) * 9 + t) * 3) * Kp + nn) * 16 + (ci % 16);
        w3[base] = (unsigned short)(hu >> 16);
        w3[base + (size_t)Kp * 16] = (unsigned short)(mu >> 16);
        w3[base + (size_t)2 * Kp * 16] = (unsigned short)(lu >> 16);
      }
      if (n->d_w3_init) { hipFree(n->d_w3_init); n->d_w3_init = nullptr; }
      AGZ_HIP_TRY(hipMalloc(&n->d_w3_init, w3.size() * 2));
      AGZ_HIP_TRY(hipMemcpyAsync(n->d_w3_init, w3.data(), w3.size() * 2, hipMemcpyHostToDevice, s));
      AGZ_HIP_TRY(hipStreamSynchronize(s));
    }
    pi += 3; bi++;
  }
  // --- dual blocks: n ordered per block tile [a-channels | b-channels]
  for (auto& p : n->d_w_dual) if (p) hipFree(p);
  for (auto& p : n->d_ep_dual) if (p) hipFree(p);
  n->d_w_dual.assign(c.SharedLayers, nullptr);
  n->d_ep_dual.assign(c.SharedLayers, nullptr);
  for (auto& p : n->d_w3_dual) if (p) hipFree(p);
  n->d_w3_dual.assign(c.SharedLayers, nullptr);
  for (auto& p : n->d_w2_dual) if (p) hipFree(p);
  n->d_w2_dual.assign(c.SharedLayers, nullptr);
  n->w_unscale.assign(c.SharedLayers, 1.0f);
  for (auto& p : n->d_u3_dual) if (p) hipFree(p);
  n->d_u3_dual.clear();

  const int half = (n->cfg == 0) ? 64 : 32;  // channels per block tile (BNT/2)
  for (int l = 0; l < c.SharedLayers; l++) {
    const Param& wa = n->params[pi];
    const Param& wb = n->params[pi + 3];
    std::vector<float> wt((size_t)9 * 2 * Kp * Kp, 0.f);
    for (int br = 0; br < 2; br++) {
      const Param& w = br == 0 ? wa : wb;
      for (int o = 0; o < K; o++) {
        int tile = o / half, j = o % half;
        int nidx = tile * 2 * half + br * half + j;
        for (int ci = 0; ci < K; ci++) for (int t = 0; t < 9; t++)
          wt[((size_t)t * 2 * Kp + nidx) * Kp + ci] = w.v[((size_t)o * K + ci) * 9 + t];
      }
    }
    std::vector<float> ep((size_t)hw * Kp * 4, 0.f);
    for (int p = 0; p < hw; p++) for (int o = 0; o < K; o++) {
      float* e = &ep[((size_t)p * Kp + o) * 4];
      fold(bi, o, n->params[pi + 1], n->params[pi + 2], p, &e[0], &e[1]);
      fold(bi + 1, o, n->params[pi + 4], n->params[pi + 5], p, &e[2], &e[3]);
    }
    int r;
    if ((r = upload(&n->d_w_dual[l], wt, s)) != AGZ_OK) return r;
    if ((r = upload(&n->d_ep_dual[l], ep, s)) != AGZ_OK) return r;
    if ((int)n->h_ep_dual.size() != c.SharedLayers) n->h_ep_dual.assign(c.SharedLayers, std::vector<float>());
    n->h_ep_dual[l] = ep;
    if (n->cfg == 0) {
      // bf16x3 image of the same (tile-interleaved) filter: w3[cc16][tap][piece][n][16], exact truncation split
      const int NC16 = Kp / 16, Ntot = 2 * Kp;
      std::vector<unsigned short> w3((size_t)9 * NC16 * 3 * Ntot * 16);
      for (int t = 0; t < 9; t++) for (int nn = 0; nn < Ntot; nn++) for (int ci = 0; ci < Kp; ci++) {
        float v = wt[((size_t)t * Ntot + nn) * Kp + ci];
        uint32_t u, hu, mu, lu; memcpy(&u, &v, 4);
        hu = u & 0xffff0000u; float hf; memcpy(&hf, &hu, 4);
        float r1 = v - hf; uint32_t ru; memcpy(&ru, &r1, 4);
        mu = ru & 0xffff0000u; float mf; memcpy(&mf, &mu, 4);
        float r2 = r1 - mf; memcpy(&lu, &r2, 4);
        size_t base = (((size_t)((ci / 16) * 9 + t) * 3) * Ntot + nn) * 16 + (ci % 16);
        w3[base] = (unsigned short)(hu >> 16);
        w3[base + (size_t)Ntot * 16] = (unsigned short)(mu >> 16);
        w3[base + (size_t)2 * Ntot * 16] = (unsigned short)(lu >> 16);
      }
      // fp16x2 image: w2[cc32][tap][piece][n][32], scaled by a power of two so that max|w| lands in [2^13, 2^14)
      {
        const int NC32 = Kp / 32;
        float wmax = 0.f;
        for (float v : wt) wmax = std::max(wmax, std::fabs(v));
        int ex = 0;
        if (wmax > 0.f) std::frexp(wmax, &ex);             // wmax = f * 2^ex, f in [0.5, 1)
        const float sb = wmax > 0.f ? std::ldexp(1.0f, 14 - ex) : 1.0f;
        n->w_unscale[l] = 1.0f / sb;
        std::vector<_Float16> w2((size_t)9 * NC32 * 2 * Ntot * 32);
        for (int t = 0; t < 9; t++) for (int nn = 0; nn < Ntot; nn++) for (int ci = 0; ci < Kp; ci++) {
          float xs = wt[((size_t)t * Ntot + nn) * Kp + ci] * sb;
          _Float16 hi = (_Float16)xs;
          _Float16 lo = (_Float16)(xs - (float)hi);
          size_t base = (((size_t)((ci / 32) * 9 + t) * 2) * Ntot + nn) * 32 + (ci % 32);
          w2[base] = hi;
          w2[base + (size_t)Ntot * 32] = lo;
        }
        if (n->d_w2_dual[l]) { hipFree(n->d_w2_dual[l]); n->d_w2_dual[l] = nullptr; }
        AGZ_HIP_TRY(hipMalloc(&n->d_w2_dual[l], w2.size() * 2));
        AGZ_HIP_TRY(hipMemcpyAsync(n->d_w2_dual[l], w2.data(), w2.size() * 2, hipMemcpyHostToDevice, s));
        AGZ_HIP_TRY(hipStreamSynchronize(s));
      }
      if (n->d_w3_dual[l]) { hipFree(n->d_w3_dual[l]); n->d_w3_dual[l] = nullptr; }
      AGZ_HIP_TRY(hipMalloc(&n->d_w3_dual[l], w3.size() * 2));
      AGZ_HIP_TRY(hipMemcpyAsync(n->d_w3_dual[l], w3.data(), w3.size() * 2, hipMemcpyHostToDevice, s));
      AGZ_HIP_TRY(hipStreamSynchronize(s));
      // latency-regime fp16x2 image (conv_lat.hpp): rows equilibrated by t_in[ci], columns by su[n], hi / lo fp16, w2[cc32][tap][piece][n][32]
      if (agz::conv_lat_ok(Kp, Kp, n->Wp)) {
        if ((int)n->d_lat_w2.size() != c.SharedLayers) {
          n->d_lat_w2.assign(c.SharedLayers, nullptr); n->d_lat_tin.assign(c.SharedLayers, nullptr); n->d_lat_colun.assign(c.SharedLayers, nullptr);
        }
        std::vector<float> tin(Kp, 1.0f), colun(Ntot, 1.0f), su(Ntot, 1.0f), rmax(Kp, 0.f), cmax(Ntot, 0.f);
        for (int t = 0; t < 9; t++) for (int nn = 0; nn < Ntot; nn++) for (int ci = 0; ci < Kp; ci++)
          rmax[ci] = std::max(rmax[ci], std::fabs(wt[((size_t)t * Ntot + nn) * Kp + ci]));
        for (int ci = 0; ci < Kp; ci++)
          if (rmax[ci] > 0.f && std::isfinite(rmax[ci])) { int ex = 0; std::frexp(rmax[ci], &ex); tin[ci] = std::ldexp(1.0f, std::max(-100, std::min(100, ex - 1))); }
        for (int t = 0; t < 9; t++) for (int nn = 0; nn < Ntot; nn++) for (int ci = 0; ci < Kp; ci++)
          cmax[nn] = std::max(cmax[nn], std::fabs(wt[((size_t)t * Ntot + nn) * Kp + ci] / tin[ci]));
        for (int nn = 0; nn < Ntot; nn++)
          if (cmax[nn] > 0.f && std::isfinite(cmax[nn])) { int ex = 0; std::frexp(cmax[nn], &ex); su[nn] = std::ldexp(1.0f, 14 - ex); colun[nn] = 1.0f / su[nn]; }
        std::vector<_Float16> l2((size_t)9 * (Kp / 32) * 2 * Ntot * 32);
        for (int t = 0; t < 9; t++) for (int nn = 0; nn < Ntot; nn++) for (int ci = 0; ci < Kp; ci++) {
          const float xs = wt[((size_t)t * Ntot + nn) * Kp + ci] / tin[ci] * su[nn];   // exact: powers of two
          const _Float16 hi = (_Float16)xs;
          const _Float16 lo = (_Float16)(xs - (float)hi);
          const size_t base = (((size_t)((ci / 32) * 9 + t) * 2) * Ntot + nn) * 32 + (ci % 32);
          l2[base] = hi;
          l2[base + (size_t)Ntot * 32] = lo;
        }
        if (n->d_lat_w2[l]) { hipFree(n->d_lat_w2[l]); n->d_lat_w2[l] = nullptr; }
        AGZ_HIP_TRY(hipMalloc(&n->d_lat_w2[l], l2.size() * 2));
        AGZ_HIP_TRY(hipMemcpyAsync(n->d_lat_w2[l], l2.data(), l2.size() * 2, hipMemcpyHostToDevice, s));
        int r2;
        if ((r2 = upload(&n->d_lat_tin[l], tin, s)) != AGZ_OK) return r2;
        if ((r2 = upload(&n->d_lat_colun[l], colun, s)) != AGZ_OK) return r2;
        AGZ_HIP_TRY(hipStreamSynchronize(s));
      }
    }
    pi += 6; bi += 2;
  }
  // --- heads
  {
    std::vector<float> hc((size_t)3 * Kp, 0.f), hb((size_t)3 * hw * 2, 0.f);
    const Param& pw = n->params[pi];  // PolicyHead filter [2,K,1,1]
    for (int o = 0; o < 2; o++) for (int ci = 0; ci < K; ci++) hc[(size_t)o * Kp + ci] = pw.v[(size_t)o * K + ci];
    for (int o = 0; o < 2; o++) for (int p = 0; p < hw; p++)
      fold(bi, o, n->params[pi + 1], n->params[pi + 2], p, &hb[((size_t)o * hw + p) * 2], &hb[((size_t)o * hw + p) * 2 + 1]);
    pi += 3; bi++;
    int r;
    if ((r = upload(&n->d_Wp, n->params[pi].v, s)) != AGZ_OK) return r;
    if ((r = upload(&n->d_bp, n->params[pi + 1].v, s)) != AGZ_OK) return r;
    pi += 2;
    const Param& vw = n->params[pi];  // ValueHead filter [1,K,1,1]
    for (int ci = 0; ci < K; ci++) hc[(size_t)2 * Kp + ci] = vw.v[ci];
    for (int p = 0; p < hw; p++)
      fold(bi, 0, n->params[pi + 1], n->params[pi + 2], p, &hb[((size_t)2 * hw + p) * 2], &hb[((size_t)2 * hw + p) * 2 + 1]);
    pi += 3; bi++;
    if ((r = upload(&n->d_W1, n->params[pi].v, s)) != AGZ_OK) return r;
    if ((r = upload(&n->d_b1, n->params[pi + 1].v, s)) != AGZ_OK) return r;
    pi += 2;
    if ((r = upload(&n->d_W2, n->params[pi].v, s)) != AGZ_OK) return r;
    if ((r = upload(&n->d_b2, n->params[pi + 1].v, s)) != AGZ_OK) return r;
    pi += 2;
    if ((r = upload(&n->d_head_conv, hc, s)) != AGZ_OK) return r;
    if ((r = upload(&n->d_head_bn, hb, s)) != AGZ_OK) return r;
    (void)A; (void)FCn;
  }
  n->committed = true;
  for (auto& p : n->d_u2_dual) if (p) hipFree(p);
  n->d_u2_dual.clear();
  for (auto& p : n->d_u2_tin) if (p) hipFree(p);
  n->d_u2_tin.clear();
  for (auto& p : n->d_u2_colun) if (p) hipFree(p);
  n->d_u2_colun.clear();
  n->free_u2c();
  for (auto& p : n->d_ep_h2) if (p) hipFree(p);
  n->d_ep_h2.clear();
  if (n->d_ep_init_h2) { hipFree(n->d_ep_init_h2); n->d_ep_init_h2 = nullptr; }
  if (n->compute_mode == AGZ_COMPUTE_WINO && n->cfg == 0) return n->build_wino_weights();
  if ((n->compute_mode == AGZ_COMPUTE_WINO_H2 || n->compute_mode == AGZ_COMPUTE_AUTO) && n->cfg == 0) return n->build_wino_h2_weights();
  return AGZ_OK;
}

int agz_wino_h2_tile(int H, int W) { return agz::wino_h2_pick_tm(H, W); }

int agz_wino_h2_chained(int H, int W, int K) {
  static const int form_env = [] { const char* e = getenv("AGZ_WINO_H2_FORM"); return e ? atoi(e) : -1; }();
  const int Kp = agz::round_up(K, 32);
  return form_env != 0 && agz::wino_h2c_ok(H, W, agz::wino_h2_pick_tm(H, W), Kp) ? 1 : 0;
}

int agz_net_set_wino_h2_form(agz_net* n, int form) {
  AGZ_REQUIRE(n, AGZ_E_INVALID, "agz_net_set_wino_h2_form: null net");
  AGZ_REQUIRE(form >= -1 && form <= 2, AGZ_E_INVALID, "agz_net_set_wino_h2_form: form %d (want -1, 0 or 1)", form);
  n->wino_form = form;
  return AGZ_OK;
}

int agz_net_min_same_batch(agz_net* n, int k, int G, int* batch) {
  AGZ_REQUIRE(n && batch && k >= 1 && G >= k, AGZ_E_INVALID, "agz_net_min_same_batch: bad argument");
  AGZ_REQUIRE(n->committed, AGZ_E_STATE, "agz_net_min_same_batch: commit the net first");
  *batch = n->min_same_batch(k, G);
  return AGZ_OK;
}

int agz_net_set_wino_h2_gemm(agz_net* n, int variant) {
  AGZ_REQUIRE(n, AGZ_E_INVALID, "agz_net_set_wino_h2_gemm: null net");
  AGZ_REQUIRE(((variant & 63) <= 2 && (variant >> 6) <= 1) || ((variant & 15) == 2 && (variant >> 4) <= 3), AGZ_E_INVALID, "agz_net_set_wino_h2_gemm: variant %d (want 0, 1, 2, + 64, or 2 + 16 * mode)", variant);
  n->wino_gemm = variant;
  return AGZ_OK;
}

int agz_wino_stages(agz_ctx* ctx, const float* x, const float* w, int B, int H, int W, int C, int N, float* V, float* M) {
  AGZ_REQUIRE(ctx && x && w && V && M, AGZ_E_INVALID, "agz_wino_stages: NULL argument");
  AGZ_REQUIRE(B >= 1 && H >= 1 && W >= 1 && C >= 16 && C % 16 == 0 && N >= 1, AGZ_E_INVALID, "agz_wino_stages: bad shape");
  AGZ_HIP_TRY(hipSetDevice(ctx->device));
  const int Hp = H + 2, Wp = W + 2;
  std::vector<float> xp((size_t)B * Hp * Wp * C, 0.f);
  for (int b = 0; b < B; b++) for (int h = 0; h < H; h++)
    memcpy(&xp[(((size_t)b * Hp + h + 1) * Wp + 1) * C], &x[((size_t)b * H + h) * W * C], (size_t)W * C * sizeof(float));
  std::vector<unsigned short> u3;
  agz::wino_build_u3(u3, N, C, [&](int n, int ci, int tap) -> double { return (double)w[((size_t)n * C + ci) * 9 + tap]; });
  agz::WinoArgs a{};
  a.B = B; a.H = H; a.W = W; a.Hp = Hp; a.Wp = Wp; a.C = C; a.Cout_p = N; a.Ntot = N;
  const size_t T = (size_t)B * agz::ceil_div(H, 4) * agz::ceil_div(W, 4);
  AGZ_REQUIRE((size_t)36 * T * C * 4 < ((size_t)1 << 32), AGZ_E_UNSUPPORTED, "agz_wino_stages: V above 4 GiB");
  float *dx = nullptr, *dV = nullptr, *dM = nullptr;
  unsigned short* dU = nullptr;
  int rc = AGZ_OK;
  if (hipMalloc(&dx, xp.size() * 4) != hipSuccess || hipMalloc(&dV, (size_t)36 * T * C * 4) != hipSuccess ||
      hipMalloc(&dM, (size_t)36 * T * N * 4) != hipSuccess || hipMalloc(&dU, u3.size() * 2) != hipSuccess) {
    agz::set_error("agz_wino_stages: out of device memory");
    rc = AGZ_E_NOMEM;
  } else {
    hipStream_t s = ctx->stream;
    bool ok = hipMemcpyAsync(dx, xp.data(), xp.size() * 4, hipMemcpyHostToDevice, s) == hipSuccess &&
              hipMemcpyAsync(dU, u3.data(), u3.size() * 2, hipMemcpyHostToDevice, s) == hipSuccess;
    a.x = dx; a.V = dV; a.U3 = dU; a.Mb = dM;
    if (ok) agz::wino_launch(ctx, a, agz::WINO_IN | agz::WINO_GEMM);
    ok = ok && hipGetLastError() == hipSuccess &&
         hipMemcpyAsync(V, dV, (size_t)36 * T * C * 4, hipMemcpyDeviceToHost, s) == hipSuccess &&
         hipMemcpyAsync(M, dM, (size_t)36 * T * N * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    if (!ok) { agz::set_error("agz_wino_stages: HIP failure: %s", hipGetErrorString(hipGetLastError())); rc = AGZ_E_HIP; }
  }
  if (dx) hipFree(dx);
  if (dV) hipFree(dV);
  if (dM) hipFree(dM);
  if (dU) hipFree(dU);
  return rc;
}

int agz_net_set_compute_mode(agz_net* n, int mode) {
  AGZ_REQUIRE(n, AGZ_E_INVALID, "agz_net_set_compute_mode: NULL net");
  const int base = mode & ~AGZ_COMPUTE_FORCE;
  AGZ_REQUIRE(base == AGZ_COMPUTE_F32_MFMA || base == AGZ_COMPUTE_BF16X3 || base == AGZ_COMPUTE_FP16X2 || base == AGZ_COMPUTE_WINO || base == AGZ_COMPUTE_AUTO ||
                  base == AGZ_COMPUTE_WINO_H2,
              AGZ_E_INVALID, "agz_net_set_compute_mode: unknown mode %d", mode);
  n->compute_mode = base;
  n->compute_force = (mode & AGZ_COMPUTE_FORCE) != 0;
  if (base == AGZ_COMPUTE_WINO && n->committed && n->cfg == 0 && n->d_u3_dual.empty()) return n->build_wino_weights();
  if ((base == AGZ_COMPUTE_WINO_H2 || base == AGZ_COMPUTE_AUTO) && n->committed && n->cfg == 0 && n->d_u2_dual.empty()) return n->build_wino_h2_weights();
  return AGZ_OK;
}

int agz_net_set_tower_queues(agz_net* n, int queues) {
  AGZ_REQUIRE(n && queues >= 0 && queues <= 2, AGZ_E_INVALID, "agz_net_set_tower_queues: queues must be 0 (auto), 1 or 2");
  n->tower_queues = queues;
  return AGZ_OK;
}

int agz_net_set_latency_mode(agz_net* n, int on) {
  AGZ_REQUIRE(n, AGZ_E_INVALID, "agz_net_set_latency_mode: NULL net");
  n->latency_mode = on != 0;
  return AGZ_OK;
}

int agz_net_infer_dev(agz_net* n, const float* planes_dev, int B, float* policy_dev, float* value_dev) {
  AGZ_REQUIRE(n && planes_dev && policy_dev && value_dev && B >= 1, AGZ_E_INVALID, "agz_net_infer_dev: bad argument");
  AGZ_HIP_TRY(hipSetDevice(n->ctx->device));
  return n->forward_dev(planes_dev, B, policy_dev, value_dev);
}

int agz_net_infer(agz_net* n, const float* planes, int B, float* policy, float* value) {
  AGZ_REQUIRE(n && planes && policy && value && B >= 1, AGZ_E_INVALID, "agz_net_infer: bad argument");
  AGZ_REQUIRE(n->committed, AGZ_E_STATE, "agz_net_infer: call agz_net_commit first");
  AGZ_HIP_TRY(hipSetDevice(n->ctx->device));
  int r = n->ensure_batch(B);
  if (r != AGZ_OK) return r;
  hipStream_t s = n->ctx->stream;
  size_t per = (size_t)n->conf.Features * n->HW;
  AGZ_HIP_TRY(hipMemcpyAsync(n->d_planes, planes, (size_t)B * per * sizeof(float), hipMemcpyHostToDevice, s));
  r = n->forward_dev(n->d_planes, B, n->d_policy, n->d_value);
  if (r != AGZ_OK) return r;
  AGZ_HIP_TRY(hipMemcpyAsync(policy, n->d_policy, (size_t)B * n->conf.ActionSpace * sizeof(float), hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipMemcpyAsync(value, n->d_value, (size_t)B * sizeof(float), hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipStreamSynchronize(s));
  return AGZ_OK;
}

int agz_net_save(const agz_net* n, const char* path) {
  AGZ_REQUIRE(n && path, AGZ_E_INVALID, "agz_net_save: NULL argument");
  FILE* f = fopen(path, "wb");
  AGZ_REQUIRE(f, AGZ_E_INVALID, "agz_net_save: cannot open %s", path);
  bool ok = fwrite("AGZNET01", 1, 8, f) == 8 && fwrite(&n->conf, sizeof(agz_net_conf), 1, f) == 1;
  uint64_t np = n->params.size();
  ok = ok && fwrite(&np, 8, 1, f) == 1;
  for (const Param& p : n->params) {
    uint64_t cnt = p.v.size();
    ok = ok && fwrite(&cnt, 8, 1, f) == 1 && fwrite(p.v.data(), 4, cnt, f) == cnt;
  }
  for (const BNStats& b : n->bn) {
    uint64_t C = b.mean.size();
    ok = ok && fwrite(&C, 8, 1, f) == 1 && fwrite(b.mean.data(), 4, C, f) == C && fwrite(b.var.data(), 4, C, f) == C;
  }
  ok = (fclose(f) == 0) && ok;
  AGZ_REQUIRE(ok, AGZ_E_INVALID, "agz_net_save: write to %s failed", path);
  return AGZ_OK;
}

int agz_net_load(agz_net* n, const char* path) {
  AGZ_REQUIRE(n && path, AGZ_E_INVALID, "agz_net_load: NULL argument");
  FILE* f = fopen(path, "rb");
  AGZ_REQUIRE(f, AGZ_E_INVALID, "agz_net_load: cannot open %s", path);
  char magic[8];
  agz_net_conf c;
  uint64_t np = 0;
  bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "AGZNET01", 8) == 0 && fread(&c, sizeof(c), 1, f) == 1 && fread(&np, 8, 1, f) == 1;
  if (ok) ok = memcmp(&c, &n->conf, sizeof(c)) == 0 && np == n->params.size();
  if (!ok) { fclose(f); agz::set_error("agz_net_load: %s is not a checkpoint of this network configuration", path); return AGZ_E_INVALID; }
  for (Param& p : n->params) {
    uint64_t cnt = 0;
    ok = ok && fread(&cnt, 8, 1, f) == 1 && cnt == p.v.size() && fread(p.v.data(), 4, cnt, f) == cnt;
  }
  for (BNStats& b : n->bn) {
    uint64_t C = 0;
    ok = ok && fread(&C, 8, 1, f) == 1 && C == b.mean.size() && fread(b.mean.data(), 4, C, f) == C && fread(b.var.data(), 4, C, f) == C;
  }
  fclose(f);
  AGZ_REQUIRE(ok, AGZ_E_INVALID, "agz_net_load: %s is truncated or mismatched", path);
  n->committed = false;
  return agz_net_commit(n);
}

double agz_net_flops_per_eval(const agz_net* n) {
  if (!n) return 0;
  double K = n->conf.K, F = n->conf.Features, hw = n->HW, A = n->conf.ActionSpace, L = n->conf.SharedLayers, FCn = n->conf.FC;
  return 2 * F * K * 9 * hw + L * 2 * (2 * K * K * 9 * hw) + (2 * K * 2 * hw + 2 * 2 * hw * A) + (2 * K * hw + 2 * hw * FCn + 2 * FCn);
}

}  // extern "C"
