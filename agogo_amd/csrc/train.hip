// dual.Train on device: training-mode forward, backward, vanilla SGD.
//
// Reference: dualnet/meta.go:16-54 (Train), dualnet/dual.go:50-132 (fwd + bwd graph),
// dualnet/ermahagerdmonards.go:106-147 (the "xent" on logits), G.NewVanillaSolver(lr 0.1) meta.go:17-20.
// Learnables keep the reference's FULL shapes: BN gamma/beta [B,C,H,W] and FC biases [B,units] are batch-shaped
// (SURVEY App. B b3/b5); inference uses row 0 (agz_trainer_export == the copy loop of dual.Infer, meta.go:141-146).
//
// Device layout: activations / conv outputs padded NHWC (zero halo) like the inference tower; the two convs of a
// block run as ONE GEMM with 2K output channels [a | b]; gamma/beta stored [B][HW][C] so they line up with the
// GEMM rows.  All learnables live in one flat buffer P and all gradients in one flat buffer G (same offsets):
// the SGD step is a single axpy and a data-parallel run needs ONE all-reduce over G (agz_trainer_grads_dev).
//
// Kernels, per compute mode (agz_trainer_set_compute_mode; every mode inside the same gradient tolerance against oracle/train.hpp):
//   AGZ_COMPUTE_F32_MFMA  forward and data-gradient convolutions through conv3x3_mfma_kernel (raw epilogue; the data gradient is the
//                         same GEMM with tap-flipped, transposed weights), weight gradient k_wgrad (fp32 MFMA)            125 ms / G19 step
//   AGZ_COMPUTE_BF16X3    the three GEMMs on the bf16 pipe with exact three-way operand splits (conv3x3_x3 raw, k_wgrad_x3)    82 ms
//   AGZ_COMPUTE_WINO_H2   forward = the DIRECT fp16x2 convolution as a pure DMA GEMM (k_conv_h2dma below: fp16 hi / lo planes of the layer
//                         input — written by the BatchNorm pass that produces the tensor, shared with the weight gradient — and the
//                         fp16x2 weight image, both by LDS-DMA; 0.55-0.58 ms per layer), data gradient = the Winograd fp16x2 path
//                         (conv_wino_h2.hpp), weight gradient = k_wgrad_h2t3 (hi / lo fp16 planes of both operands, LDS-DMA, three taps
//                         per workgroup from one x image, ds_read_b64_tr_b16): 0.55 ms per layer alone, 0.48 of the fp16x2 MFMA roof;
//                         every layer's weight images are built at the start of the step on the side stream (prep_weights)  42.5-43.3 ms
// BatchNorm statistics / apply / backward are bandwidth-bound elementwise + reduction kernels at ~5 TB/s (43 % of a layer's time: the
// reference's batch-shaped gamma / beta fix their bytes); the heads run as wave-per-row reductions and LDS-tiled FC kernels (second form,
// 0.35 ms per step); the batch-shaped gamma / beta take their SGD step inside k_bn_bwd1 on single-process steps; the weight gradient
// runs on its own stream; a data-parallel step reduces every layer's slice of the flat gradient buffer under the backward pass
// (on_slice, comm.hip).
#include <cmath>
#include <cstring>
#include <thread>
#include <functional>
#include <vector>

#include "net.hpp"
#include "conv_maps.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace agz {

struct TGeo { int B, H, W, HW, Hp, Wp, M; };  // M = B*HW rows

__host__ __device__ constexpr __forceinline__ size_t pix_off(const TGeo& g, int r) {  // padded pixel index of GEMM row r
  int b = r / g.HW, p = r - b * g.HW;
  int h = p / g.W, w = p - h * g.W;
  return ((size_t)b * g.Hp + h + 1) * g.Wp + w + 1;
}
// (the same function as conv_maps.hpp's pix, which the CPU check of k_conv_h2dma3's image geometry uses: compared at compile time)
constexpr bool pix_off_is_cmaps_pix(int B, int H, int W) {
  const TGeo g{B, H, W, H * W, H + 2, W + 2, B * H * W};
  for (int m = 0; m < g.M; m++)
    if (pix_off(g, m) != cmaps::pix(m, g.HW, g.W, g.Hp, g.Wp)) return false;
  return true;
}
static_assert(pix_off_is_cmaps_pix(3, 19, 19) && pix_off_is_cmaps_pix(2, 16, 17) && pix_off_is_cmaps_pix(5, 3, 4), "pix_off != cmaps::pix");

// ---- BatchNorm statistics: per-channel sums over the M interior rows of z [pix][C] -----------------------------
// pass 0: sum(z)                    -> acc[c]
// pass 1: sum((z - mean)^2)         -> acc[c]      (two-pass variance like the oracle)
__global__ void k_bn_sum(TGeo g, const float* __restrict__ z, int C, const float* __restrict__ mean, double* __restrict__ acc,
                         int rows_per_block) {
  int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, g.M);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float mu = mean ? mean[c] : 0.f;
    double s = 0;
    for (int r = r0; r < r1; r++) {
      float v = z[pix_off(g, r) * C + c];
      if (mean) { float d = v - mu; s += (double)d * d; } else s += v;
    }
    atomicAdd(&acc[c], s);
  }
}
// finalize: pass 0 -> mean = acc/m ; pass 1 -> inv = 1/sqrt(acc/m + eps).  Clears acc.
__global__ void k_bn_fin(double* acc, int C, double m, float eps, float* mean, float* inv, int pass) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  if (pass == 0) mean[c] = (float)(acc[c] / m); else inv[c] = 1.0f / sqrtf((float)(acc[c] / m) + eps);
  acc[c] = 0;
}

// One pass for both statistics (C % 4 == 0, C <= 1024): sum(z) -> acc[c], sum(z^2) -> acc[1024 + c], both in double (a product of
// two fp32 values is exact in double, so E[z^2] - mean^2 loses nothing against the two-pass form above at these row counts); float4
// loads, 256 / (C/4) row slices per workgroup reduced through LDS, one double atomic per channel and workgroup.
__global__ __launch_bounds__(256) void k_bn_stats(TGeo g, const float* __restrict__ z, int C, double* __restrict__ acc, int rows_per_block) {
  __shared__ double red[2][1024];
  const int c4n = C >> 2, slices = 256 / c4n;
  const int tid = threadIdx.x, cg = tid % c4n, rs = tid / c4n;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, g.M);
  double s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (rs < slices) {
    int r = r0 + rs;
    for (; r + 3 * slices < r1; r += 4 * slices) {   // four rows in flight (one load per iteration left the kernel at 2.4 TB/s)
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const float4*>(z + pix_off(g, r + u * slices) * C + 4 * cg);
#pragma unroll
      for (int u = 0; u < 4; u++) {
        s[0] += v[u].x; s[1] += v[u].y; s[2] += v[u].z; s[3] += v[u].w;
        q[0] += (double)v[u].x * v[u].x; q[1] += (double)v[u].y * v[u].y; q[2] += (double)v[u].z * v[u].z; q[3] += (double)v[u].w * v[u].w;
      }
    }
    for (; r < r1; r += slices) {
      const float4 v = *reinterpret_cast<const float4*>(z + pix_off(g, r) * C + 4 * cg);
      s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
      q[0] += (double)v.x * v.x; q[1] += (double)v.y * v.y; q[2] += (double)v.z * v.z; q[3] += (double)v.w * v.w;
    }
  }
  for (int c = tid; c < C; c += 256) { red[0][c] = 0; red[1][c] = 0; }
  __syncthreads();
  for (int k = 0; k < slices; k++) {          // slice after slice: a fixed summation order inside the workgroup
    if (rs == k) {
#pragma unroll
      for (int e = 0; e < 4; e++) { red[0][4 * cg + e] += s[e]; red[1][4 * cg + e] += q[e]; }
    }
    __syncthreads();
  }
  for (int c = tid; c < C; c += 256) { atomicAdd(&acc[c], red[0][c]); atomicAdd(&acc[1024 + c], red[1][c]); }
}
__global__ void k_bn_fin2(double* acc, int C, double m, float eps, float* mean, float* inv) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double mu = acc[c] / m;
  double var = acc[1024 + c] / m - mu * mu;
  var = var > 0 ? var : 0;
  mean[c] = (float)mu;
  inv[c] = 1.0f / sqrtf((float)var + eps);
  acc[c] = 0; acc[1024 + c] = 0;
}

// ---- tower BN apply (+ReLU, + dual add + ReLU).  z [pix][nbr*Kp]; gamma/beta [M][nbr*Kp]; out [pix][Kp] --------
__global__ void k_bn_apply(TGeo g, const float* __restrict__ z, const float* __restrict__ gamma, const float* __restrict__ beta,
                           const float* __restrict__ mean, const float* __restrict__ inv, float* __restrict__ out, int Kp, int nbr) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)g.M * Kp) return;
  int r = (int)(idx / Kp), c = (int)(idx - (size_t)r * Kp);
  size_t po = pix_off(g, r);
  int C = nbr * Kp;
  float acc = 0.f;
  for (int br = 0; br < nbr; br++) {
    int cc = br * Kp + c;
    float xh = (z[po * C + cc] - mean[cc]) * inv[cc];
    float y = gamma[(size_t)r * C + cc] * xh + beta[(size_t)r * C + cc];
    acc += y > 0.f ? y : 0.f;
  }
  out[po * Kp + c] = (nbr == 2) ? (acc > 0.f ? acc : 0.f) : acc;
}

// block-wide maximum of |.| bit patterns -> one atomicMax per block (the range word of the fp16x2 weight gradient's split)
__device__ __forceinline__ void block_amax_commit(unsigned m, unsigned* __restrict__ out_bits) {
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o, 64); m = t > m ? t : m; }
  __shared__ unsigned sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned a = sm[0] > sm[1] ? sm[0] : sm[1], b = sm[2] > sm[3] ? sm[2] : sm[3];
    atomicMax(out_bits, a > b ? a : b);
  }
}
// per-board maxima of a workgroup's rows (rows_per_block <= HW: at most two boards, bf and bf + 1) -> one atomicMax per board and block:
// the per-board range words the fp16x2 convolutions scale their input rows by (board_amax_parts_kernel swept the stored tensor for them)
__device__ __forceinline__ void board_amax_commit(unsigned m0, unsigned m1, int bf, int B, unsigned* __restrict__ board_bits) {
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned t0 = (unsigned)__shfl_xor((int)m0, o, 64), t1 = (unsigned)__shfl_xor((int)m1, o, 64);
    m0 = t0 > m0 ? t0 : m0; m1 = t1 > m1 ? t1 : m1;
  }
  __shared__ unsigned sb[2][4];
  if ((threadIdx.x & 63) == 0) { sb[0][threadIdx.x >> 6] = m0; sb[1][threadIdx.x >> 6] = m1; }
  __syncthreads();
  if (threadIdx.x < 2 && bf + (int)threadIdx.x < B) {
    const unsigned* q = sb[threadIdx.x];
    const unsigned a = q[0] > q[1] ? q[0] : q[1], b = q[2] > q[3] ? q[2] : q[3];
    atomicMax(&board_bits[bf + threadIdx.x], a > b ? a : b);
  }
}
// The same apply, four channels per thread and a fixed channel quad per thread (Kp % 4 == 0, (Kp / 4) divides 256), with max|out| of
// the whole tensor as a by-product: the NEXT layer's weight gradient splits this tensor into fp16 pieces and needs its range — a
// separate sweep (k_absmax) re-read it.  Same expressions per element as k_bn_apply.
// ph != nullptr (round 5): the fp16 hi / lo planes of `out` that the NEXT layer's DMA convolution and weight gradient read are written
// here as well, scaled by the power of two of `est_bits` — the exact range of the same tensor one step earlier.  The exact range of THIS
// step comes out of this very pass; k_split_h2p_cond then keeps the planes if both ranges have the same exponent (the usual case: the
// scale is what the exact range prescribes, the results do not depend on the history) and re-splits the tensor otherwise.
__device__ __forceinline__ float wg_h2_scale(unsigned amax_bits);
typedef _Float16 bn_f16x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_bn_apply_v(TGeo g, const float* __restrict__ z, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ mean,
                                                    const float* __restrict__ inv, float* __restrict__ out, int Kp, int nbr,
                                                    int rows_per_block, unsigned* __restrict__ amax_bits, unsigned* __restrict__ board_bits,
                                                    _Float16* __restrict__ ph = nullptr, _Float16* __restrict__ pl = nullptr, const unsigned* __restrict__ est_bits = nullptr) {
  const int q = Kp >> 2, tpr = 256 / q, cq = threadIdx.x % q, rs = threadIdx.x / q;
  const float ps = ph ? wg_h2_scale(*est_bits) : 1.f;
  const int C = nbr * Kp;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, g.M);
  float4 mu[2], iv[2];
  for (int br = 0; br < nbr; br++) {
    mu[br] = reinterpret_cast<const float4*>(mean + br * Kp)[cq];
    iv[br] = reinterpret_cast<const float4*>(inv + br * Kp)[cq];
  }
  unsigned mx = 0, mb[2] = {0, 0};
  const int bf = r0 / g.HW, r_next = (bf + 1) * g.HW;   // (board_bits: rows_per_block <= HW, so rows >= r_next belong to board bf + 1)
  for (int r = r0 + rs; r < r1; r += tpr) {
    const size_t po = pix_off(g, r);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int br = 0; br < nbr; br++) {
      const float4 zz = reinterpret_cast<const float4*>(z + po * C + br * Kp)[cq];
      const float4 gm = reinterpret_cast<const float4*>(gamma + (size_t)r * C + br * Kp)[cq];
      const float4 bt = reinterpret_cast<const float4*>(beta + (size_t)r * C + br * Kp)[cq];
      const float zv[4] = {zz.x, zz.y, zz.z, zz.w}, gv[4] = {gm.x, gm.y, gm.z, gm.w}, bv[4] = {bt.x, bt.y, bt.z, bt.w};
      const float mv[4] = {mu[br].x, mu[br].y, mu[br].z, mu[br].w}, nv[4] = {iv[br].x, iv[br].y, iv[br].z, iv[br].w};
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float xh = (zv[k] - mv[k]) * nv[k];
        const float y = gv[k] * xh + bv[k];
        acc[k] += y > 0.f ? y : 0.f;
      }
    }
    float o[4];
    unsigned rowm = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      o[k] = (nbr == 2) ? (acc[k] > 0.f ? acc[k] : 0.f) : acc[k];
      const unsigned b = __float_as_uint(o[k]) & 0x7fffffffu;
      rowm = b > rowm ? b : rowm;
    }
    mx = rowm > mx ? rowm : mx;
    if (r >= r_next) mb[1] = rowm > mb[1] ? rowm : mb[1]; else mb[0] = rowm > mb[0] ? rowm : mb[0];
    reinterpret_cast<float4*>(out + po * Kp)[cq] = make_float4(o[0], o[1], o[2], o[3]);
    if (ph) {
      bn_f16x4_t hh, ll;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const float xs = o[k] * ps;
        hh[k] = (_Float16)xs;
        ll[k] = (_Float16)(xs - (float)hh[k]);
      }
      reinterpret_cast<bn_f16x4_t*>(ph + po * Kp)[cq] = hh;
      reinterpret_cast<bn_f16x4_t*>(pl + po * Kp)[cq] = ll;
    }
  }
  block_amax_commit(mx, amax_bits);
  if (board_bits) board_amax_commit(mb[0], mb[1], bf, g.B, board_bits);
}

// ---- tower BN backward, step 1: d(out) -> dgamma, dbeta, d(xhat) (stored in dz) and the two channel sums -------
// fuse_lr != 0 (single-process steps, agz_trainer_batch / agz_train_dev): the batch-shaped gamma / beta — 98 % of the learnables,
// 7.7 GB at G19 — take their SGD step HERE (w -= lr * grad on the value just read) instead of writing the gradient and having
// agz_trainer_apply sweep parameters and gradients again: 23 GB less traffic per step.  Same arithmetic (one fp32 multiply-subtract per
// element, as k_axpy); the gradients of these tensors are then not materialised (agz_trainer_forward_backward keeps them: the
// data-parallel path reduces them before its apply).
__global__ void k_bn_bwd1(TGeo g, const float* __restrict__ z, float* gamma, float* beta,
                          const float* __restrict__ mean, const float* __restrict__ inv, const float* __restrict__ out,
                          const float* __restrict__ dout, float* __restrict__ dgamma, float* __restrict__ dbeta,
                          float* __restrict__ dz, double* __restrict__ s1, double* __restrict__ s2, int Kp, int nbr,
                          int rows_per_block, float fuse_lr) {
  int C = nbr * Kp;
  int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, g.M);
  for (int cc = threadIdx.x; cc < C; cc += blockDim.x) {
    int c = cc % Kp;
    double a1 = 0, a2 = 0;
    for (int r = r0; r < r1; r++) {
      size_t po = pix_off(g, r);
      float g0 = dout[po * Kp + c];
      if (nbr == 2 && !(out[po * Kp + c] > 0.f)) g0 = 0.f;
      float xh = (z[po * C + cc] - mean[cc]) * inv[cc];
      float gm = gamma[(size_t)r * C + cc];
      float bt = beta[(size_t)r * C + cc];
      float y = gm * xh + bt;
      float gg = y > 0.f ? g0 : 0.f;
      if (fuse_lr != 0.f) {   // (uniform) p += alpha * g with alpha = -lr, exactly k_axpy's expression
        gamma[(size_t)r * C + cc] = gm + (-fuse_lr) * (gg * xh);
        beta[(size_t)r * C + cc] = bt + (-fuse_lr) * gg;
      } else {
        dgamma[(size_t)r * C + cc] = gg * xh;
        dbeta[(size_t)r * C + cc] = gg;
      }
      float dxh = gg * gm;
      dz[po * C + cc] = dxh;
      a1 += dxh; a2 += (double)dxh * xh;
    }
    atomicAdd(&s1[cc], a1);
    atomicAdd(&s2[cc], a2);
  }
}
// step 2: dz = inv * (dxhat - s1/m - xhat*s2/m)
__global__ void k_bn_bwd2(TGeo g, const float* __restrict__ z, const float* __restrict__ mean, const float* __restrict__ inv,
                          float* __restrict__ dz, const double* __restrict__ s1, const double* __restrict__ s2, int C) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)g.M * C) return;
  int r = (int)(idx / C), cc = (int)(idx - (size_t)r * C);
  size_t po = pix_off(g, r);
  float xh = (z[po * C + cc] - mean[cc]) * inv[cc];
  float m = (float)g.M;
  dz[po * C + cc] = inv[cc] * (dz[po * C + cc] - (float)s1[cc] / m - xh * ((float)s2[cc] / m));
}

// step 2, four channels per thread, a fixed channel quad per thread (C % 4 == 0, (C / 4) divides 256: the per-channel terms are
// loaded once), with max|dz| of the tensor as a by-product (the weight gradient's range word).  Same expression per element.
__global__ __launch_bounds__(256) void k_bn_bwd2_v(TGeo g, const float* __restrict__ z, const float* __restrict__ mean,
                                                   const float* __restrict__ inv, float* __restrict__ dz, const double* __restrict__ s1,
                                                   const double* __restrict__ s2, int C, int rows_per_block, unsigned* __restrict__ amax_bits,
                                                   unsigned* __restrict__ board_bits) {
  const int q = C >> 2, tpr = 256 / q, cq = threadIdx.x % q, rs = threadIdx.x / q;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, g.M);
  const float m = (float)g.M;
  float mv[4], nv[4], a1[4], a2[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int cc = 4 * cq + k;
    mv[k] = mean[cc]; nv[k] = inv[cc]; a1[k] = (float)s1[cc] / m; a2[k] = (float)s2[cc] / m;
  }
  unsigned mx = 0, mb[2] = {0, 0};
  const int bf = r0 / g.HW, r_next = (bf + 1) * g.HW;
  for (int r = r0 + rs; r < r1; r += tpr) {
    const size_t po = pix_off(g, r);
    const float4 zz = reinterpret_cast<const float4*>(z + po * C)[cq];
    const float4 dd = reinterpret_cast<const float4*>(dz + po * C)[cq];
    const float zv[4] = {zz.x, zz.y, zz.z, zz.w}, dv[4] = {dd.x, dd.y, dd.z, dd.w};
    float o[4];
    unsigned rowm = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const float xh = (zv[k] - mv[k]) * nv[k];
      o[k] = nv[k] * (dv[k] - a1[k] - xh * a2[k]);
      const unsigned b = __float_as_uint(o[k]) & 0x7fffffffu;
      rowm = b > rowm ? b : rowm;
    }
    mx = rowm > mx ? rowm : mx;
    if (r >= r_next) mb[1] = rowm > mb[1] ? rowm : mb[1]; else mb[0] = rowm > mb[0] ? rowm : mb[0];
    reinterpret_cast<float4*>(dz + po * C)[cq] = make_float4(o[0], o[1], o[2], o[3]);
  }
  block_amax_commit(mx, amax_bits);
  if (board_bits) board_amax_commit(mb[0], mb[1], bf, g.B, board_bits);
}

// ---- weight gradient: dW[tap][n][c] += sum_r dz[pix(r)][n] * x[pix(r)+off(tap)][c]  (fp32 MFMA, split over rows) ----
// block tile 128 (n) x 128 (c), 4 waves x (2x2 MFMA 32x32x2), K-loop over `rows` GEMM rows in steps of 32; partial
// results are added with float atomics (one wgrad launch per layer; order-dependent rounding is within tolerance).
struct WgArgs {
  const float* dz; const float* x; float* dw;
  TGeo g; int N, Cin, rows_per_block, n_tiles, c_tiles, n_chunks;
};
__global__ __launch_bounds__(256) void k_wgrad(WgArgs a) {
  __shared__ float As[32][132];  // [k row][n]   (+4 pad: conflict-free ds_read_b32 across the two lane halves)
  __shared__ float Bs[32][132];  // [k row][c]
  int bid = blockIdx.x;
  int ct = bid % a.c_tiles; bid /= a.c_tiles;
  int nt = bid % a.n_tiles; bid /= a.n_tiles;
  int tap = bid % 9; int chunk = bid / 9;
  int n0 = nt * 128, c0 = ct * 128;
  int ky = tap / 3, kx = tap - ky * 3;
  long tapoff = (long)(ky - 1) * a.g.Wp + (kx - 1);
  int r_begin = chunk * a.rows_per_block, r_end = min(r_begin + a.rows_per_block, a.g.M);
  int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wm = wid >> 1, wn = wid & 1;
  f32x16 acc[2][2];
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  // staging: 32 rows x 128 cols = 1024 float4 / 256 threads = 4 each
  for (int rb = r_begin; rb < r_end; rb += 32) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      int e = tid + 256 * q;        // float4 index
      int row = e >> 5, col = (e & 31) * 4;
      int r = rb + row;
      float4 va = make_float4(0, 0, 0, 0), vb = va;
      if (r < r_end) {
        size_t po = pix_off(a.g, r);
        if (n0 + col < a.N) va = *reinterpret_cast<const float4*>(a.dz + po * a.N + n0 + col);
        if (c0 + col < a.Cin) vb = *reinterpret_cast<const float4*>(a.x + (size_t)((long)po + tapoff) * a.Cin + c0 + col);
      }
      *reinterpret_cast<float4*>(&As[row][col]) = va;
      *reinterpret_cast<float4*>(&Bs[row][col]) = vb;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < 32; k += 2) {
      int kr = k + (lane >> 5);
      float a0 = As[kr][wm * 64 + (lane & 31)], a1 = As[kr][wm * 64 + 32 + (lane & 31)];
      float b0 = Bs[kr][wn * 64 + (lane & 31)], b1 = Bs[kr][wn * 64 + 32 + (lane & 31)];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        int n = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int c = c0 + wn * 64 + j * 32 + (lane & 31);
        if (n < a.N && c < a.Cin) atomicAdd(&a.dw[((size_t)tap * a.N + n) * a.Cin + c], acc[i][j][r]);
      }
}

// ---- the same weight gradient on the bf16 matrix pipe (AGZ_COMPUTE_BF16X3) ------------------------------------------------
// Both GEMM operands are activations here (dz and x), so both are split on the way into LDS: every fp32 value = three bf16
// pieces by exact truncation (conv_x3.hpp), six v_mfma_f32_32x32x16_bf16 per product instead of sixteen fp32-MFMA passes.  The
// reduction dimension is the pixel row r, and the MFMA wants eight consecutive k per lane: a thread fetches the SAME column of
// eight consecutive rows (a wave = 64 consecutive columns of one row per load: 256-byte runs), splits, and writes the eight
// bf16 of one piece as one 16-byte LDS word.  LDS image per operand and piece: [128 columns][32 k] bf16, the 16-byte slot of
// (column, k group) XOR-ed with bits 2..3 of the column: ds_write_b128 (16 lanes = 16 columns, one k group) and the fragment
// ds_read_b128 (16 lanes = 16 columns) both touch 16 different slots of the 256-byte bank window.
// Tile 128 (n) x 128 (c), 4 waves of 64 x 64, K step 32 rows; next step's 32 values per thread are fetched under the MFMAs.
typedef short wg_bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned wg_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void wg_split(float v, unsigned& h, unsigned& m, unsigned& l) {
  unsigned hu = __float_as_uint(v) & 0xffff0000u;
  float r = v - __uint_as_float(hu);
  unsigned mu = __float_as_uint(r) & 0xffff0000u;
  float r2 = r - __uint_as_float(mu);
  h = hu; m = mu; l = __float_as_uint(r2);
}
__device__ __forceinline__ unsigned wg_pack(unsigned e0, unsigned e1) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }
__device__ __forceinline__ unsigned wg_lds_off(int col, int kg) { return (unsigned)(col * 64 + ((kg ^ ((col >> 2) & 3)) << 4)); }

__global__ __launch_bounds__(256, 2) void k_wgrad_x3(WgArgs a) {
  constexpr int PIECE = 128 * 64;                 // one piece image of one operand: 128 columns x 32 k x 2 B
  __shared__ __attribute__((aligned(16))) unsigned char lds[6 * PIECE];   // A (dz) pieces 0..2, B (x) pieces 3..5
  // XCD-aware order: the 9 taps x tiles of one row chunk re-read the same dz / x rows (6 MB per 2048 rows at K = 256); workgroup ids
  // go round-robin over the 8 XCDs, so chunk c runs entirely on XCD c % 8 and its re-reads hit that XCD's L2 instead of missing in 8
  const int per_chunk = a.n_tiles * a.c_tiles * 9;
  const int local = (int)(blockIdx.x >> 3);
  const int chunk = (local / per_chunk) * 8 + (int)(blockIdx.x & 7);
  if (chunk >= a.n_chunks) return;
  int bid = local % per_chunk;
  const int ct = bid % a.c_tiles; bid /= a.c_tiles;
  const int nt = bid % a.n_tiles; bid /= a.n_tiles;
  const int tap = bid;
  const int n0 = nt * 128, c0 = ct * 128;
  const int ky = tap / 3, kx = tap - ky * 3;
  const long tapoff = (long)(ky - 1) * a.g.Wp + (kx - 1);
  const int r_begin = chunk * a.rows_per_block, r_end = min(r_begin + a.rows_per_block, a.g.M);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wid >> 1, wn = wid & 1;
  const int kg = wid;                             // staging: this wave's k group = rows rb + 8 kg .. + 7
  const bool na0 = n0 + lane < a.N, na1 = n0 + 64 + lane < a.N;
  const bool cb0 = c0 + lane < a.Cin, cb1 = c0 + 64 + lane < a.Cin;
  // (buffer loads, masked lanes and rows past the chunk read zero through an out-of-range offset: see k_wgrad_h2)
  const size_t n_pix = (size_t)a.g.B * a.g.Hp * a.g.Wp;
  const __amdgpu_buffer_rsrc_t rdz = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz), 0, (int)(n_pix * a.N * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)(n_pix * a.Cin * 4), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  float va0[8], va1[8], vb0[8], vb1[8];
  auto fetch = [&](int rb) {
    const int r0 = rb + kg * 8;                   // wave-uniform
    int b = r0 / a.g.HW, p = r0 - b * a.g.HW;
    int h = p / a.g.W, w = p - h * a.g.W;
#pragma unroll
    for (int q = 0; q < 8; q++) {
      // (rows past the last board lie past the end of both buffers and read zero; chunk ends are multiples of the 32-row K step)
      const int po = (b * a.g.Hp + h + 1) * a.g.Wp + w + 1;
      const unsigned oa = (unsigned)(po * a.N + n0 + lane) * 4u;
      const unsigned ob = (unsigned)((po + (int)tapoff) * a.Cin + c0 + lane) * 4u;
      va0[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdz, na0 ? oa : OOB, 0, 0));
      va1[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rdz, na1 ? oa + 256u : OOB, 0, 0));
      vb0[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, cb0 ? ob : OOB, 0, 0));
      vb1[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rx, cb1 ? ob + 256u : OOB, 0, 0));
      const bool we = ++w == a.g.W;                 // (selects, not branches)
      w = we ? 0 : w; h += we ? 1 : 0;
      const bool he = h == a.g.H;
      h = he ? 0 : h; b += he ? 1 : 0;
    }
  };
  auto stage = [&](const float* v, int piece0, int col) {   // split eight k of one column, one 16-byte word per piece
    unsigned hh[8], mm[8], ll[8];
#pragma unroll
    for (int q = 0; q < 8; q++) wg_split(v[q], hh[q], mm[q], ll[q]);
    wg_u32x4_t ph, pm, pl;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      ph[q] = wg_pack(hh[2 * q], hh[2 * q + 1]);
      pm[q] = wg_pack(mm[2 * q], mm[2 * q + 1]);
      pl[q] = wg_pack(ll[2 * q], ll[2 * q + 1]);
    }
    const unsigned off = wg_lds_off(col, kg);
    *reinterpret_cast<wg_u32x4_t*>(lds + (piece0 + 0) * PIECE + off) = ph;
    *reinterpret_cast<wg_u32x4_t*>(lds + (piece0 + 1) * PIECE + off) = pm;
    *reinterpret_cast<wg_u32x4_t*>(lds + (piece0 + 2) * PIECE + off) = pl;
  };
  fetch(r_begin);
  for (int rb = r_begin; rb < r_end; rb += 32) {
    stage(va0, 0, lane); stage(va1, 0, 64 + lane);
    stage(vb0, 3, lane); stage(vb1, 3, 64 + lane);
    __syncthreads();
    if (rb + 32 < r_end) fetch(rb + 32);          // in flight under the MFMAs below
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      const int kgr = ks * 2 + (lane >> 5);
      wg_bf16x8_t A_[2][3], B_[2][3];
#pragma unroll
      for (int pz = 0; pz < 3; pz++) {
#pragma unroll
        for (int i = 0; i < 2; i++)
          A_[i][pz] = *reinterpret_cast<const wg_bf16x8_t*>(lds + pz * PIECE + wg_lds_off(wm * 64 + i * 32 + (lane & 31), kgr));
#pragma unroll
        for (int j = 0; j < 2; j++)
          B_[j][pz] = *reinterpret_cast<const wg_bf16x8_t*>(lds + (3 + pz) * PIECE + wg_lds_off(wn * 64 + j * 32 + (lane & 31), kgr));
      }
#define WG_MF(I, J, PA, PB) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_[I][PA], B_[J][PB], acc[I][J], 0, 0, 0);
#define WG_QUAD(PA, PB) WG_MF(0, 0, PA, PB) WG_MF(0, 1, PA, PB) WG_MF(1, 0, PA, PB) WG_MF(1, 1, PA, PB)
      WG_QUAD(2, 0) WG_QUAD(0, 2) WG_QUAD(1, 1) WG_QUAD(1, 0) WG_QUAD(0, 1) WG_QUAD(0, 0)   // smallest terms first
#undef WG_QUAD
#undef WG_MF
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        int n = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int c = c0 + wn * 64 + j * 32 + (lane & 31);
        if (n < a.N && c < a.Cin) atomicAdd(&a.dw[((size_t)tap * a.N + n) * a.Cin + c], acc[i][j][r]);
      }
}

// ---- the weight gradient with fp16x2 products (AGZ_COMPUTE_WINO_H2) --------------------------------------------------------
// k_wgrad_x3 spends as many cycles splitting as multiplying: every element of dz and x is split (4 VALU ops + packing) by each of the
// 9 tap workgroups that read it, and a product costs six MFMAs.  Here both operands are split ONCE per layer by an elementwise pass
// (k_absmax -> power-of-two scale putting the tensor's maximum into [2^13, 2^14); k_split_h2 writes hi = RN16(v s), lo = RN16(v s - hi)
// as one 32-bit word per element, 4 bytes like the fp32 it replaces), the GEMM's staging only regroups eight rows of a column into
// 16-byte words (one v_perm per two halves), and a product is three v_mfma_f32_32x32x16_f16 (hi hi, hi lo, lo hi; the dropped
// lo lo <= 2^-22 relative).  hi + lo carries v s to an absolute error <= 2^-25 (scaled units) = 2^-38 of the tensor's range.
// The sums come out scaled by s_dz s_x and are un-scaled (exact powers of two) before the atomic add.
typedef _Float16 wg_f16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float wg_h2_scale(unsigned amax_bits) {   // 2^(13 - E(amax)); 1 for an all-zero (or non-finite) tensor
  const int e = (int)((amax_bits >> 23) & 0xff);
  if (e == 0 || e == 255) return 1.f;
  int se = 127 + 13 - (e - 127);
  se = se < 1 ? 1 : (se > 254 ? 254 : se);
  return __uint_as_float((unsigned)se << 23);
}
__global__ __launch_bounds__(256) void k_absmax(const float* __restrict__ x, size_t n4, unsigned* __restrict__ out_bits) {
  unsigned m = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const unsigned a = __float_as_uint(v.x) & 0x7fffffffu, b = __float_as_uint(v.y) & 0x7fffffffu;
    const unsigned c = __float_as_uint(v.z) & 0x7fffffffu, d = __float_as_uint(v.w) & 0x7fffffffu;
    const unsigned ab = a > b ? a : b, cd = c > d ? c : d, q = ab > cd ? ab : cd;
    m = q > m ? q : m;
  }
  for (int o = 32; o > 0; o >>= 1) { const unsigned t = (unsigned)__shfl_xor((int)m, o, 64); m = t > m ? t : m; }
  __shared__ unsigned sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned a = sm[0] > sm[1] ? sm[0] : sm[1], b = sm[2] > sm[3] ? sm[2] : sm[3];
    atomicMax(out_bits, a > b ? a : b);
  }
}
__global__ __launch_bounds__(256) void k_split_h2(const float* __restrict__ x, unsigned* __restrict__ y, size_t n4, const unsigned* __restrict__ amax_bits) {
  const float s = wg_h2_scale(*amax_bits);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float in[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
    unsigned o[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const _Float16 hi = (_Float16)in[k];
      const _Float16 lo = (_Float16)(in[k] - (float)hi);
      o[k] = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
    }
    reinterpret_cast<uint4*>(y)[i] = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
struct WgH2Args {
  const unsigned* dz2; const unsigned* x2;   // (hi | lo << 16) per element, the layouts of dz and x
  const unsigned* amax;                      // [0] dz, [1] x: bits of max|.|
  float* dw;
  TGeo g; int N, Cin, rows_per_block, n_tiles, c_tiles, n_chunks;
};
// LDS image of this kernel: [k group of 8 rows (4)][column slot (128)][16 B = 8 k], slot = column with its low two bits XORed by
// bits 4-5 (the staging writes of a half wave — columns 4 cg + c, 8 bytes each — then fall on 16 distinct 16-byte bank groups, and the
// fragment reads of 32 consecutive columns stay a permutation of 32 consecutive slots)
__device__ __forceinline__ unsigned wg2_lds_off(int col, int kg) { return (unsigned)(kg * 2048 + ((col ^ ((col >> 4) & 3)) << 4)); }
typedef unsigned wg_u32x2_t __attribute__((ext_vector_type(2)));
// Loads are 16 bytes (four channels of one row), 8 per thread and step, the next step's issued under this step's MFMAs.  Deeper
// prefetch (2-4 steps in flight in registers) measured no faster, 4 workgroups per CU (this form: 123 registers) 0.67 ms per G19
// layer against 0.76 for round 3's kernel (profiles/r04/wgrad_ab.log): the bound is the L2 -> CU operand stream (see k_wgrad_h2t3, which
// replaces this kernel on boards >= 16 wide).
__global__ __launch_bounds__(256, 2) void k_wgrad_h2(WgH2Args a) {
  constexpr int D = 1;                             // K steps in flight in registers (2, 3, 4 measured no faster: the comment above)
  constexpr int PIECE = 128 * 64;                 // one piece image of one operand: 128 columns x 32 k x 2 B
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * PIECE];   // A (dz) hi, lo; B (x) hi, lo
  // XCD-aware order: the 9 taps x tiles of one row chunk re-read the same dz / x rows (6 MB per 2048 rows at K = 256); workgroup ids
  // go round-robin over the 8 XCDs, so chunk c runs entirely on XCD c % 8 and its re-reads hit that XCD's L2 instead of missing in 8
  const int per_chunk = a.n_tiles * a.c_tiles * 9;
  const int local = (int)(blockIdx.x >> 3);
  const int chunk = (local / per_chunk) * 8 + (int)(blockIdx.x & 7);
  if (chunk >= a.n_chunks) return;
  int bid = local % per_chunk;
  const int ct = bid % a.c_tiles; bid /= a.c_tiles;
  const int nt = bid % a.n_tiles; bid /= a.n_tiles;
  const int tap = bid;
  const int n0 = nt * 128, c0 = ct * 128;
  const int ky = tap / 3, kx = tap - ky * 3;
  const int tapoff = (ky - 1) * a.g.Wp + (kx - 1);
  const int r_begin = chunk * a.rows_per_block, r_end = min(r_begin + a.rows_per_block, a.g.M);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wid >> 1, wn = wid & 1;
  const int cg = lane & 31, half = lane >> 5;     // staging: columns 4 cg .. + 3, rows 8 wid + 4 half .. + 3 of the step
  const bool na = n0 + 4 * cg < a.N, cb = c0 + 4 * cg < a.Cin;   // (N and Cin are multiples of 4)
  // buffer loads with a per-lane byte offset: masked lanes and rows past the chunk get an out-of-range offset and read zero — a
  // conditional global_load costs a saveexec + two branches per load
  const size_t n_pix = (size_t)a.g.B * a.g.Hp * a.g.Wp;
  const __amdgpu_buffer_rsrc_t rdz = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(a.dz2), 0, (int)(n_pix * a.N * 4), 0x00020000);
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(a.x2), 0, (int)(n_pix * a.Cin * 4), 0x00020000);
  constexpr unsigned OOB = 0x80000000u;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  wg_u32x4_t va[D][4], vb[D][4];
  auto fetch = [&](wg_u32x4_t* pa, wg_u32x4_t* pb, int rb) {
    const int r0 = rb + wid * 8;                  // wave-uniform
    int b = r0 / a.g.HW, p = r0 - b * a.g.HW;
    int h = p / a.g.W, w = p - h * a.g.W;
    int r = r0 + 4 * half;
#pragma unroll
    for (int q = 0; q < 4; q++) {                 // the upper half wave starts four rows on (selects, not branches)
      const bool adv = half != 0;
      const bool we = adv && (w + 1 == a.g.W);
      w = adv ? (we ? 0 : w + 1) : w; h += we ? 1 : 0;
      const bool he = h == a.g.H;
      h = he ? 0 : h; b += he ? 1 : 0;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      // (rows past the last board lie past the end of both buffers and read zero; rows of the padding steps past the chunk are masked)
      const int po = (b * a.g.Hp + h + 1) * a.g.Wp + w + 1;
      const bool in = r + q < r_end;
      const unsigned oa = (unsigned)(po * a.N + n0 + 4 * cg) * 4u;
      const unsigned ob = (unsigned)((po + tapoff) * a.Cin + c0 + 4 * cg) * 4u;
      pa[q] = __builtin_bit_cast(wg_u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rdz, (na && in) ? oa : OOB, 0, 0));
      pb[q] = __builtin_bit_cast(wg_u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, (cb && in) ? ob : OOB, 0, 0));
      const bool we = ++w == a.g.W;
      w = we ? 0 : w; h += we ? 1 : 0;
      const bool he = h == a.g.H;
      h = he ? 0 : h; b += he ? 1 : 0;
    }
  };
  auto stage = [&](const wg_u32x4_t* v, int piece0) {   // four k of four columns: per column one 8-byte word of hi halves, one of lo halves
#pragma unroll
    for (int c = 0; c < 4; c++) {
      wg_u32x2_t ph, pl;
      ph[0] = __builtin_amdgcn_perm(v[1][c], v[0][c], 0x05040100u);
      ph[1] = __builtin_amdgcn_perm(v[3][c], v[2][c], 0x05040100u);
      pl[0] = __builtin_amdgcn_perm(v[1][c], v[0][c], 0x07060302u);
      pl[1] = __builtin_amdgcn_perm(v[3][c], v[2][c], 0x07060302u);
      const unsigned off = wg2_lds_off(4 * cg + c, wid) + 8u * half;
      *reinterpret_cast<wg_u32x2_t*>(lds + (piece0 + 0) * PIECE + off) = ph;
      *reinterpret_cast<wg_u32x2_t*>(lds + (piece0 + 1) * PIECE + off) = pl;
    }
  };
#pragma unroll
  for (int d = 0; d < D; d++) {
    fetch(va[d], vb[d], r_begin + 32 * d);
    __builtin_amdgcn_sched_barrier(0);            // issue order = consumption order: the loop's waits count loads, whichever edge it is entered by
  }
  // (the step count is rounded up to a multiple of D: the padding steps multiply zeros)
  for (int rb = r_begin; rb < r_end; rb += 32 * D) {
#pragma unroll
    for (int d = 0; d < D; d++) {
      stage(va[d], 0); stage(vb[d], 2);
      __syncthreads();
      fetch(va[d], vb[d], rb + 32 * (d + D));     // in flight under the next D steps
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        const int kgr = ks * 2 + (lane >> 5);
        wg_f16x8_t A_[2][2], B_[2][2];
#pragma unroll
        for (int pz = 0; pz < 2; pz++) {
#pragma unroll
          for (int i = 0; i < 2; i++)
            A_[i][pz] = *reinterpret_cast<const wg_f16x8_t*>(lds + pz * PIECE + wg2_lds_off(wm * 64 + i * 32 + (lane & 31), kgr));
#pragma unroll
          for (int j = 0; j < 2; j++)
            B_[j][pz] = *reinterpret_cast<const wg_f16x8_t*>(lds + (2 + pz) * PIECE + wg2_lds_off(wn * 64 + j * 32 + (lane & 31), kgr));
        }
#define WG_MF(I, J, PA, PB) acc[I][J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[I][PA], B_[J][PB], acc[I][J], 0, 0, 0);
#define WG_QUAD(PA, PB) WG_MF(0, 0, PA, PB) WG_MF(0, 1, PA, PB) WG_MF(1, 0, PA, PB) WG_MF(1, 1, PA, PB)
        WG_QUAD(1, 0) WG_QUAD(0, 1) WG_QUAD(0, 0)   // smaller terms first
#undef WG_QUAD
#undef WG_MF
      }
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);          // (the next step's regrouping hoisted into these MFMAs would wait on the NEWEST loads)
    }
  }
  const float un = 1.f / (wg_h2_scale(a.amax[0]) * wg_h2_scale(a.amax[1]));   // exact: a power of two
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        int n = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int c = c0 + wn * 64 + j * 32 + (lane & 31);
        if (n < a.N && c < a.Cin) atomicAdd(&a.dw[((size_t)tap * a.N + n) * a.Cin + c], acc[i][j][r] * un);
      }
}

// ---- the weight gradient with the operands DMA'd into LDS and transposed by the LDS read (gfx950: ds_read_b64_tr_b16) ---------------
// What bounds the register-staged kernel above is the L2 -> CU operand stream: 32 KB per workgroup and K step for 128 x 128 x 32
// products is 6.9 GB per G19 layer, 10.3 TB/s at 0.67 ms — the rate round 3's l2_probe and round 4's all-from-cache GEMM floor gave
// for that path (a single-tap DMA form of the same tile measured 0.88 ms: the same bytes through the narrower DMA path).  Fewer
// bytes per product: the three taps of one kernel ROW read the same dz rows and x rows one pixel apart.  k_split_h2p writes the hi
// and lo halves as two fp16 planes in the operand's own [pixel][channel] layout; a plane's rows arrive by buffer_load ... lds
// (16 bytes per lane: one row, eight channels; each lane computes ITS row's padded pixel, so shifts and masks cost nothing) into
// the image [row / 4][channel / 8 (16)][row % 4][16 B] — a DMA instruction fills 1 KB (four rows) — and are read k-major by the
// transposing LDS read: within 16 lanes, lane i supplies the address of (row i / 4, channels 4 (i % 4) .. + 3) and receives channel
// i's four rows (scripts/probes/tr16_probe.hip confirms the lane mapping); the 32 lanes such a read serves per cycle touch 256
// contiguous bytes.  The x image is indexed by PADDED PIXEL (40 consecutive pixels cover a step's 32 rows, its <= 2 row ends and the
// +-1 shifts): one landing serves all three taps, row of (k, kx) = delta(k) + kx, delta(k) = k + 2 (row ends between the step's first
// row and row k).  36 KB per step for 3 x the products: 2.6 x less traffic per product.  K steps never straddle a board (12 steps of 32
// rows per 19 x 19 board, the last one 9 rows + zeros: 6 % padding).  Boards >= 16 wide (two row ends per step at most); others keep
// the register-staged kernel.  G19 layer: 0.55 ms (0.76 in round 3); without the DMA 0.41, DMA alone 0.34, neither 0.12 (the
// epilogue's float atomics): landing and arithmetic of two workgroups per CU overlap only partly (profiles/r04/wgrad_ab.log).
typedef short wg_s4_t __attribute__((vector_size(8)));
typedef _Float16 wg_f16x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* wg_lds_ptr_t;
typedef unsigned wg_rsrc_t __attribute__((ext_vector_type(4)));
// One LDS-DMA instruction, written out: the compiler's own bookkeeping of buffer_load ... lds keeps at most eight such stores apart and
// waits vmcnt(0) before LDS reads it can no longer tell from the NEXT stage's DMA (seen in the ISA: a drain in the middle of every K
// step).  As inline assembly the DMA is invisible to that pass; the waits are the counted ones below.  (Compiler-known vector-memory
// instructions stay safe: its counts can only be too strict, never too loose, with extra loads in flight.)
__device__ __forceinline__ void wg_dma16(wg_rsrc_t rsrc, unsigned lds_addr, unsigned voff) {
  unsigned keep;                                   // (M0 is put back: the compiler does not model assembly writes to it)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc) : "memory");
}
typedef __attribute__((address_space(3))) wg_s4_t* wg_lds_s4_t;
__global__ __launch_bounds__(256) void k_split_h2p(const float* __restrict__ x, _Float16* __restrict__ yh, _Float16* __restrict__ yl, size_t n4,
                                                   const unsigned* __restrict__ amax_bits) {
  const float s = wg_h2_scale(*amax_bits);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float in[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
    wg_f16x4_t h, l;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      h[k] = (_Float16)in[k];
      l[k] = (_Float16)(in[k] - (float)h[k]);
    }
    reinterpret_cast<wg_f16x4_t*>(yh)[i] = h;
    reinterpret_cast<wg_f16x4_t*>(yl)[i] = l;
  }
}
// the same split, skipped when the planes already hold it: written by k_bn_apply_v under the previous step's range `est_bits`, valid when
// that range and this step's exact one prescribe the same power of two
__global__ __launch_bounds__(256) void k_split_h2p_cond(const float* __restrict__ x, _Float16* __restrict__ yh, _Float16* __restrict__ yl, size_t n4,
                                                        const unsigned* __restrict__ amax_bits, const unsigned* __restrict__ est_bits) {
  if (est_bits && *est_bits != 0u && wg_h2_scale(*est_bits) == wg_h2_scale(*amax_bits)) return;
  const float s = wg_h2_scale(*amax_bits);
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float in[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
    wg_f16x4_t h, l;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      h[k] = (_Float16)in[k];
      l[k] = (_Float16)(in[k] - (float)h[k]);
    }
    reinterpret_cast<wg_f16x4_t*>(yh)[i] = h;
    reinterpret_cast<wg_f16x4_t*>(yl)[i] = l;
  }
}
struct WgH2t3Args {
  const _Float16 *dzh, *dzl, *xh, *xl;
  const unsigned* amax;
  float* dw;
  TGeo g; int N, Cin, steps_per_board, n_tiles, c_tiles, n_chunks;
};
__global__ __launch_bounds__(256, 2) void k_wgrad_h2t3(WgH2t3Args a) {
  constexpr int PA = 32 * 128 * 2, PB = 40 * 128 * 2, STAGE = 2 * PA + 2 * PB;   // dz hi, lo (32 rows); x hi, lo (40 pixels)
  __shared__ __attribute__((aligned(1024))) unsigned char lds0[STAGE];
  __shared__ __attribute__((aligned(1024))) unsigned char lds1[STAGE];
  const int per_chunk = a.n_tiles * a.c_tiles * 3;   // XCD-aware order, as above
  const int local = (int)(blockIdx.x >> 3);
  const int chunk = (local / per_chunk) * 8 + (int)(blockIdx.x & 7);
  if (chunk >= a.n_chunks) return;
  int bid = local % per_chunk;
  const int ct = bid % a.c_tiles; bid /= a.c_tiles;
  const int nt = bid % a.n_tiles; bid /= a.n_tiles;
  const int ky = bid;
  const int n0 = nt * 128, c0 = ct * 128;
  const int b_begin = (int)((long)chunk * a.g.B / a.n_chunks), b_end = (int)((long)(chunk + 1) * a.g.B / a.n_chunks);   // 6 or 7 boards, say
  const int n_steps = (b_end - b_begin) * a.steps_per_board;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wid >> 1, wn = wid & 1;
  const int cs = lane >> 2, kk = lane & 3;        // DMA: this lane's eight channels and row of a four-row group
  const bool na = n0 + 8 * cs < a.N, cb = c0 + 8 * cs < a.Cin;
  const size_t n_pix = (size_t)a.g.B * a.g.Hp * a.g.Wp;
  auto mk = [](const void* p, size_t bytes) -> wg_rsrc_t {
    const unsigned long long u = (unsigned long long)p;
    wg_rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)u); r[1] = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes); r[3] = 0x00020000u;
    return r;
  };
  const wg_rsrc_t rah = mk(a.dzh, n_pix * a.N * 2), ral = mk(a.dzl, n_pix * a.N * 2);
  const wg_rsrc_t rbh = mk(a.xh, n_pix * a.Cin * 2), rbl = mk(a.xl, n_pix * a.Cin * 2);
  constexpr unsigned OOB = 0x80000000u;
  f32x16 acc[3][2][2];
#pragma unroll
  for (int t = 0; t < 3; t++)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[t][i][j][r] = 0.f;
  // step s of the chunk: board b_begin + s / steps_per_board, rows 32 (s % steps_per_board) .. of that board
  struct Step { int po0, w0, nv; };
  auto step_of = [&](int s) -> Step {
    const int bq = s / a.steps_per_board, sib = s - bq * a.steps_per_board;
    const int r0 = sib * 32, h0 = r0 / a.g.W, w0 = r0 - h0 * a.g.W;
    Step st;
    st.po0 = ((b_begin + bq) * a.g.Hp + h0 + 1) * a.g.Wp + w0 + 1;
    st.w0 = w0;
    st.nv = s < n_steps ? min(32, a.g.HW - r0) : 0;   // (a padding step past the chunk lands zeros)
    return st;
  };
  auto delta = [&](int k, int w0) -> int {         // padded-pixel distance of the step's row k from its first row (boards >= 16 wide)
    const int c = w0 + k;
    return k + (c >= a.g.W ? 2 : 0) + (c >= 2 * a.g.W ? 2 : 0);
  };
  // the landing of step s, in five pieces (two for dz: wave w lands rows 8 w .. 8 w + 7; three for x: ten four-row groups, waves 0, 1
  // land three, waves 2, 3 two) so that they can be issued BETWEEN the MFMA groups of the step before: as one block ahead of the
  // barrier their address arithmetic and issue slots (~0.4 us) delayed every step
  auto issue_piece = [&](unsigned char* stage, const Step& st, int piece) {
    const unsigned base = (unsigned)(size_t)(wg_lds_ptr_t)stage;
    if (piece < 2) {
      const int t = piece;
      const int k = 8 * wid + 4 * t + kk;
      const int po = st.po0 + delta(k, st.w0);
      const unsigned oa = (na && k < st.nv) ? (unsigned)(po * a.N + n0 + 8 * cs) * 2u : OOB;
      const unsigned d = base + (unsigned)(2 * wid + t) * 1024u;
      wg_dma16(rah, d, oa);
      wg_dma16(ral, d + PA, oa);
    } else {
      const int qb = st.po0 + (ky - 1) * a.g.Wp - 1;   // x: image row rho <-> padded pixel qb + rho (halo pixels hold zeros)
      const int m = wid + 4 * (piece - 2);
      if (m < 10) {
        const int q = qb + 4 * m + kk;
        const unsigned ob = (cb && st.nv > 0 && q >= 0) ? (unsigned)(q * a.Cin + c0 + 8 * cs) * 2u : OOB;
        const unsigned d = base + 2 * PA + (unsigned)m * 1024u;
        wg_dma16(rbh, d, ob);
        wg_dma16(rbl, d + PB, ob);
      }
    }
  };
  const int i16 = lane & 15, kh = lane >> 5;
  const unsigned cpart = (unsigned)(((((lane >> 4) & 1) * 2 + ((i16 & 3) >> 1)) * 4) * 16 + (i16 & 1) * 8);   // channel part of a read address
  const unsigned foa = (unsigned)(kh * 2048 + (i16 >> 2) * 16) + cpart;                                   // dz: rows are the k themselves
  auto tr2 = [&](const unsigned char* p0, const unsigned char* p1) -> wg_f16x8_t {
    const wg_s4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s4_t)(p0));
    const wg_s4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_lds_s4_t)(p1));
    return __builtin_shufflevector(__builtin_bit_cast(wg_f16x4_t, v0), __builtin_bit_cast(wg_f16x4_t, v1), 0, 1, 2, 3, 4, 5, 6, 7);
  };
  auto compute = [&](const unsigned char* stage, int s, unsigned char* next_stage) {
    const int w0 = step_of(s).w0;
    const Step nx = step_of(s + 1);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      wg_f16x8_t A_[2][2];
#pragma unroll
      for (int pz = 0; pz < 2; pz++)
#pragma unroll
        for (int i = 0; i < 2; i++) {
          const unsigned char* q = stage + pz * PA + foa + (wm * 64 + i * 32) * 8 + ks * 4096;
          A_[i][pz] = tr2(q, q + 1024);
        }
      const int k0 = ks * 16 + kh * 8 + (i16 >> 2);  // this lane's rows of the step: k0 and k0 + 4
      const int d0 = delta(k0, w0), d1 = delta(k0 + 4, w0);
#pragma unroll
      for (int kx = 0; kx < 3; kx++) {
        const int r0 = d0 + kx, r1 = d1 + kx;
        const unsigned o0 = (unsigned)((r0 >> 2) * 1024 + (r0 & 3) * 16) + cpart, o1 = (unsigned)((r1 >> 2) * 1024 + (r1 & 3) * 16) + cpart;
        wg_f16x8_t B_[2][2];
#pragma unroll
        for (int pz = 0; pz < 2; pz++)
#pragma unroll
          for (int j = 0; j < 2; j++) {
            const unsigned char* q = stage + 2 * PA + pz * PB + (wn * 64 + j * 32) * 8;
            B_[j][pz] = tr2(q + o0, q + o1);
          }
#define WG_MF(I, J, PA_, PB_) acc[kx][I][J] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[I][PA_], B_[J][PB_], acc[kx][I][J], 0, 0, 0);
#define WG_QUAD(PA_, PB_) WG_MF(0, 0, PA_, PB_) WG_MF(0, 1, PA_, PB_) WG_MF(1, 0, PA_, PB_) WG_MF(1, 1, PA_, PB_)
        WG_QUAD(1, 0) WG_QUAD(0, 1) WG_QUAD(0, 0)   // smaller terms first
#undef WG_QUAD
#undef WG_MF
        const int piece = ks * 3 + kx;               // next step's landing, a piece after each of the first five MFMA groups
        if (piece < 5) issue_piece(next_stage, nx, piece);
      }
    }
  };
  // raw s_barrier + explicit vmcnt: a __syncthreads() fence knows nothing of the assembly DMA
  {
    const Step st0 = step_of(0);
#pragma unroll
    for (int piece = 0; piece < 5; piece++) issue_piece(lds0, st0, piece);
  }
  for (int s = 0; s < n_steps; s += 2) {            // (an odd step count runs one step of zeros)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of step s has landed; after the barrier every wave's
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    compute(lds0, s, lds1);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                    // (also: every wave is done reading lds0 before step s + 2 lands there)
    __builtin_amdgcn_sched_barrier(0);
    compute(lds1, s + 1, lds0);
    __builtin_amdgcn_sched_barrier(0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const float un = 1.f / (wg_h2_scale(a.amax[0]) * wg_h2_scale(a.amax[1]));   // exact: a power of two
#pragma unroll
  for (int kx = 0; kx < 3; kx++)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
          int n = n0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          int c = c0 + wn * 64 + j * 32 + (lane & 31);
          if (n < a.N && c < a.Cin) atomicAdd(&a.dw[((size_t)(ky * 3 + kx) * a.N + n) * a.Cin + c], acc[kx][i][j][r] * un);
        }
}

// wt[8-tap][c][n] = wf[tap][n][c]   (data-gradient weights: flipped taps, transposed)
__global__ void k_make_wt(const float* __restrict__ wf, float* __restrict__ wt, int N, int Cin) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t tot = (size_t)9 * N * Cin;
  if (idx >= tot) return;
  int c = (int)(idx % Cin); size_t t = idx / Cin; int n = (int)(t % N); int tap = (int)(t / N);
  wt[((size_t)(8 - tap) * Cin + c) * N + n] = wf[idx];
}

__global__ void k_pack_planes_t(const float* __restrict__ planes, float* __restrict__ out, TGeo g, int F, int Fp) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)g.M * Fp) return;
  int r = (int)(idx / Fp), c = (int)(idx - (size_t)r * Fp);
  int b = r / g.HW, p = r - b * g.HW;
  out[pix_off(g, r) * Fp + c] = c < F ? planes[((size_t)b * F + c) * g.HW + p] : 0.f;
}

// dst[r][:] = src[idx[r]][:] — batch assembly for agz_train_dev (rows stay where they are; the shuffle moves indices)
__global__ void k_gather_rows_t(const float* __restrict__ src, const int32_t* __restrict__ idx, float* __restrict__ dst, int row_len,
                                size_t total) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  size_t r = i / row_len;
  dst[i] = src[(size_t)idx[r] * row_len + (i - r * row_len)];
}

__global__ void k_axpy(float* __restrict__ p, const float* __restrict__ g, float alpha, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] += alpha * g[i];
}

// ---- heads (small; one thread per output element, plain loops) --------------------------------------------------
// zh[r][j] = sum_c x[pix(r)][c] * hc[j][c], j = 0,1 policy, 2 value
__global__ void k_head_conv(TGeo g, const float* __restrict__ x, const float* __restrict__ hc, float* __restrict__ zh, int Kp) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= g.M * 3) return;
  int r = idx / 3, j = idx - r * 3;
  const float* xp = x + pix_off(g, r) * Kp;
  float s = 0.f;
  for (int c = 0; c < Kp; c++) s += xp[c] * hc[j * Kp + c];
  zh[idx] = s;
}
// per-channel stats of zh [M][3]: one block per channel (the same summation order as the single-block form: 0.28 -> 0.09 ms at G19)
__global__ void k_head_stats(TGeo g, const float* __restrict__ zh, float eps, float* __restrict__ mean, float* __restrict__ inv) {
  __shared__ double red[256];
  {
    const int j = blockIdx.x;
    double s = 0;
    for (int r = threadIdx.x; r < g.M; r += 256) s += zh[r * 3 + j];
    red[threadIdx.x] = s; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    float mu = (float)(red[0] / g.M); __syncthreads();
    s = 0;
    for (int r = threadIdx.x; r < g.M; r += 256) { float d = zh[r * 3 + j] - mu; s += (double)d * d; }
    red[threadIdx.x] = s; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) { mean[j] = mu; inv[j] = 1.0f / sqrtf((float)(red[0] / g.M) + eps); }
    __syncthreads();
  }
}
// yh[b][j][p] = relu(gamma*xhat+beta); head gamma/beta layout [B][3][HW]
__global__ void k_head_apply(TGeo g, const float* __restrict__ zh, const float* __restrict__ hg, const float* __restrict__ hb,
                             const float* __restrict__ mean, const float* __restrict__ inv, float* __restrict__ yh) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= g.M * 3) return;
  int r = idx / 3, j = idx - r * 3;
  int b = r / g.HW, p = r - b * g.HW;
  float xh = (zh[idx] - mean[j]) * inv[j];
  size_t o = ((size_t)b * 3 + j) * g.HW + p;
  float y = hg[o] * xh + hb[o];
  yh[o] = y > 0.f ? y : 0.f;
}
struct HeadT {
  int B, HW, A, FC;
  const float* yh;  // [B][3][HW]
  const float *Wp, *bp, *W1, *b1, *W2, *b2;
  const float *Pi, *V;
  float *logits, *hpre, *o;
  float *dWp, *dbp, *dW1, *db1, *dW2, *db2, *dyh;  // dyh [B][3][HW]
  float* cost;  // [2]: pcost, vcost sums
};
__global__ void k_fc_fwd(HeadT h) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  int nl = h.B * h.A, nh = h.B * h.FC;
  if (idx < nl) {
    int b = idx / h.A, j = idx - b * h.A;
    const float* yp = h.yh + (size_t)b * 3 * h.HW;  // policy features: channels 0,1 contiguous = flatten c-major
    float s = 0.f;
    for (int i = 0; i < 2 * h.HW; i++) s += yp[i] * h.Wp[(size_t)i * h.A + j];
    h.logits[idx] = s + h.bp[idx];
  } else if (idx < nl + nh) {
    int k = idx - nl;
    int b = k / h.FC, j = k - b * h.FC;
    const float* yv = h.yh + ((size_t)b * 3 + 2) * h.HW;
    float s = 0.f;
    for (int i = 0; i < h.HW; i++) s += yv[i] * h.W1[(size_t)i * h.FC + j];
    h.hpre[k] = s + h.b1[k];
  }
}
__global__ void k_value_out(HeadT h) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= h.B) return;
  float s = 0.f;
  for (int j = 0; j < h.FC; j++) { float hv = h.hpre[(size_t)b * h.FC + j]; s += (hv > 0.f ? hv : 0.f) * h.W2[j]; }
  h.o[b] = s + h.b2[b];
}
__global__ void k_cost(HeadT h) {  // single block
  __shared__ double red[256];
  double s = 0;
  for (int i = threadIdx.x; i < h.B * h.A; i += 256) s += -(double)(h.Pi[i] * h.logits[i] + (1.f - h.Pi[i]) * (1.f - h.logits[i]));
  red[threadIdx.x] = s; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) h.cost[0] = (float)(red[0] / ((double)h.B * h.A));
  __syncthreads();
  s = 0;
  for (int b = threadIdx.x; b < h.B; b += 256) { double d = h.o[b] - h.V[b]; s += d * d; }
  red[threadIdx.x] = s; __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) h.cost[1] = (float)(red[0] / h.B);
}
// backward of the FC parts.  One thread per gradient element (sums over the batch inside).
__global__ void k_fc_bwd(HeadT h) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const float sc = 1.0f / ((float)h.B * (float)h.A);
  int n_wp = 2 * h.HW * h.A, n_bp = h.B * h.A, n_w1 = h.HW * h.FC, n_b1 = h.B * h.FC, n_w2 = h.FC, n_b2 = h.B;
  int n_dy = h.B * 3 * h.HW;
  if (idx < n_wp) {  // dWp[i][j] = sum_b yp[b][i] * dl[b][j]
    int i = idx / h.A, j = idx - i * h.A;
    float s = 0.f;
    for (int b = 0; b < h.B; b++) s += h.yh[(size_t)b * 3 * h.HW + i] * ((1.f - 2.f * h.Pi[(size_t)b * h.A + j]) * sc);
    h.dWp[idx] = s;
    return;
  }
  idx -= n_wp;
  if (idx < n_bp) { h.dbp[idx] = (1.f - 2.f * h.Pi[idx]) * sc; return; }
  idx -= n_bp;
  if (idx < n_w1) {  // dW1[i][j] = sum_b yv[b][i] * dh[b][j]
    int i = idx / h.FC, j = idx - i * h.FC;
    float s = 0.f;
    for (int b = 0; b < h.B; b++) {
      float dob = 2.f * (h.o[b] - h.V[b]) / (float)h.B;
      float dh = h.hpre[(size_t)b * h.FC + j] > 0.f ? dob * h.W2[j] : 0.f;
      s += h.yh[((size_t)b * 3 + 2) * h.HW + i] * dh;
    }
    h.dW1[idx] = s;
    return;
  }
  idx -= n_w1;
  if (idx < n_b1) {
    int b = idx / h.FC, j = idx - b * h.FC;
    float dob = 2.f * (h.o[b] - h.V[b]) / (float)h.B;
    h.db1[idx] = h.hpre[idx] > 0.f ? dob * h.W2[j] : 0.f;
    return;
  }
  idx -= n_b1;
  if (idx < n_w2) {
    float s = 0.f;
    for (int b = 0; b < h.B; b++) { float hv = h.hpre[(size_t)b * h.FC + idx]; s += (2.f * (h.o[b] - h.V[b]) / (float)h.B) * (hv > 0.f ? hv : 0.f); }
    h.dW2[idx] = s;
    return;
  }
  idx -= n_w2;
  if (idx < n_b2) { h.db2[idx] = 2.f * (h.o[idx] - h.V[idx]) / (float)h.B; return; }
  idx -= n_b2;
  if (idx < n_dy) {  // d(yh)[b][j][p]
    int b = idx / (3 * h.HW), q = idx - b * 3 * h.HW;
    float s = 0.f;
    if (q < 2 * h.HW) {
      for (int j = 0; j < h.A; j++) s += ((1.f - 2.f * h.Pi[(size_t)b * h.A + j]) * sc) * h.Wp[(size_t)q * h.A + j];
    } else {
      int i = q - 2 * h.HW;
      float dob = 2.f * (h.o[b] - h.V[b]) / (float)h.B;
      for (int j = 0; j < h.FC; j++) if (h.hpre[(size_t)b * h.FC + j] > 0.f) s += dob * h.W2[j] * h.W1[(size_t)i * h.FC + j];
    }
    h.dyh[idx] = s;
  }
}
// head BN backward (3 channels, one block each): dzh[r][j], dgamma/dbeta [B][3][HW]
__global__ void k_head_bn_bwd(TGeo g, const float* __restrict__ zh, const float* __restrict__ yh, const float* __restrict__ dyh,
                              const float* __restrict__ hg, const float* __restrict__ mean, const float* __restrict__ inv,
                              float* __restrict__ dhg, float* __restrict__ dhb, float* __restrict__ dzh) {
  __shared__ double r1[256], r2[256];
  {
    const int j = blockIdx.x;
    double a1 = 0, a2 = 0;
    for (int r = threadIdx.x; r < g.M; r += 256) {
      int b = r / g.HW, p = r - b * g.HW;
      size_t o = ((size_t)b * 3 + j) * g.HW + p;
      float gg = yh[o] > 0.f ? dyh[o] : 0.f;
      float xh = (zh[r * 3 + j] - mean[j]) * inv[j];
      dhg[o] = gg * xh; dhb[o] = gg;
      float dxh = gg * hg[o];
      dzh[r * 3 + j] = dxh;
      a1 += dxh; a2 += (double)dxh * xh;
    }
    r1[threadIdx.x] = a1; r2[threadIdx.x] = a2; __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; } __syncthreads(); }
    float s1 = (float)r1[0], s2 = (float)r2[0], m = (float)g.M; __syncthreads();
    for (int r = threadIdx.x; r < g.M; r += 256) {
      float xh = (zh[r * 3 + j] - mean[j]) * inv[j];
      dzh[r * 3 + j] = inv[j] * (dzh[r * 3 + j] - s1 / m - xh * (s2 / m));
    }
    __syncthreads();
  }
}
// head conv backward: dhc[j][c] = sum_r dzh[r][j]*x[pix(r)][c] ; dx[pix(r)][c] = sum_j dzh[r][j]*hc[j][c]
__global__ void k_head_conv_bwd_w(TGeo g, const float* __restrict__ x, const float* __restrict__ dzh, float* __restrict__ dhc, int Kp,
                                  int rows_per_block) {
  int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, g.M);
  for (int e = threadIdx.x; e < 3 * Kp; e += blockDim.x) {
    int j = e / Kp, c = e - j * Kp;
    float s = 0.f;
    for (int r = r0; r < r1; r++) s += dzh[r * 3 + j] * x[pix_off(g, r) * Kp + c];
    atomicAdd(&dhc[e], s);
  }
}
__global__ void k_head_conv_bwd_x(TGeo g, const float* __restrict__ dzh, const float* __restrict__ hc, float* __restrict__ dx, int Kp) {
  size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)g.M * Kp) return;
  int r = (int)(idx / Kp), c = (int)(idx - (size_t)r * Kp);
  dx[pix_off(g, r) * Kp + c] = dzh[r * 3] * hc[c] + dzh[r * 3 + 1] * hc[Kp + c] + dzh[r * 3 + 2] * hc[2 * Kp + c];
}


// ---- heads, second form (round 5): the same arithmetic with the memory behaviour fixed.  The nine kernels above cost 1.85 ms of a 47 ms
// G19 step, serially between the forward and the backward pass: three-block BatchNorm passes over 92 416 rows, a single-block cost sum,
// one thread per output walking strided operands.  Below: wave-per-row reductions, multi-block partial sums (double atomics), and FC
// kernels that load each weight once for eight batch rows (resp. eight weight rows) from LDS tiles.  Wherever an output is a sequential sum
// over i / b / j the ORDER is the old kernel's (bit-identical outputs: the FC forward, dWp, dW1, d(yh)); the BatchNorm sums, the head
// convolution and the cost are now tree / atomic sums (differences at 1e-7 of the value).
__global__ __launch_bounds__(256) void k_head_conv2(TGeo g, const float* __restrict__ x, const float* __restrict__ hc, float* __restrict__ zh, int Kp) {
  const int lane = threadIdx.x & 63, wv = (int)((blockIdx.x * 256u + threadIdx.x) >> 6), nw = (int)(gridDim.x * 4u);
  for (int r = wv; r < g.M; r += nw) {
    const float* xp = x + pix_off(g, r) * Kp;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int c = lane * 4; c < Kp; c += 256) {
      const float4 v = *reinterpret_cast<const float4*>(xp + c);
      const float4 a = *reinterpret_cast<const float4*>(hc + c), b = *reinterpret_cast<const float4*>(hc + Kp + c), d = *reinterpret_cast<const float4*>(hc + 2 * Kp + c);
      s0 += v.x * a.x + v.y * a.y + v.z * a.z + v.w * a.w;
      s1 += v.x * b.x + v.y * b.y + v.z * b.z + v.w * b.w;
      s2 += v.x * d.x + v.y * d.y + v.z * d.z + v.w * d.w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (lane == 0) { zh[(size_t)r * 3] = s0; zh[(size_t)r * 3 + 1] = s1; zh[(size_t)r * 3 + 2] = s2; }
  }
}
// sums [3] and sums of squares [3] of zh's channels -> hacc[0..5] (zeroed by the caller), then mean / inv
__global__ __launch_bounds__(256) void k_head_stats_part(TGeo g, const float* __restrict__ zh, double* __restrict__ hacc) {
  __shared__ double red[6][4];
  double s[3] = {0, 0, 0}, q[3] = {0, 0, 0};
  for (int r = blockIdx.x * 256 + threadIdx.x; r < g.M; r += gridDim.x * 256) {
#pragma unroll
    for (int j = 0; j < 3; j++) { const double v = zh[(size_t)r * 3 + j]; s[j] += v; q[j] += v * v; }
  }
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s[j] += __shfl_xor(s[j], o, 64); q[j] += __shfl_xor(q[j], o, 64); }
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { for (int j = 0; j < 3; j++) { red[j][w] = s[j]; red[3 + j][w] = q[j]; } }
  __syncthreads();
  if (threadIdx.x < 6) atomicAdd(&hacc[threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}
__global__ void k_head_stats_fin(TGeo g, const double* __restrict__ hacc, float eps, float* __restrict__ mean, float* __restrict__ inv) {
  const int j = threadIdx.x;
  if (j >= 3) return;
  const double mu = hacc[j] / g.M;
  double var = hacc[3 + j] / g.M - mu * mu;
  if (var < 0) var = 0;
  mean[j] = (float)mu;
  inv[j] = 1.0f / sqrtf((float)var + eps);
}
// out[b][j] = sum_i y[b][i] * Wt[i][j] + bias[b][j] for eight batch rows per block: grid (ceil(N / 256), ceil(B / 8), 2); z = 0: the policy
// logits (y = channels 0, 1 of yh flattened, K = 2 HW), z = 1: the value head's hidden layer (y = channel 2, K = HW)
__global__ __launch_bounds__(256) void k_fc_fwd2(HeadT h) {
  extern __shared__ float ys[];                      // [8][K]
  const bool pol = blockIdx.z == 0;
  const int N = pol ? h.A : h.FC, K = pol ? 2 * h.HW : h.HW;
  const int j = blockIdx.x * 256 + threadIdx.x, b0 = blockIdx.y * 8;
  if (blockIdx.x * 256 >= N) return;
  const float* W = pol ? h.Wp : h.W1;
  for (int e = threadIdx.x; e < 8 * K; e += 256) {
    const int t = e / K, i = e - t * K, b = b0 + t;
    ys[e] = b < h.B ? h.yh[((size_t)b * 3 + (pol ? 0 : 2)) * h.HW + i] : 0.f;
  }
  __syncthreads();
  if (j >= N) return;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int i = 0; i < K; i++) {
    const float w = W[(size_t)i * N + j];
#pragma unroll
    for (int t = 0; t < 8; t++) acc[t] += ys[t * K + i] * w;
  }
#pragma unroll
  for (int t = 0; t < 8; t++) {
    const int b = b0 + t;
    if (b >= h.B) break;
    const size_t o = (size_t)b * N + j;
    if (pol) h.logits[o] = acc[t] + h.bp[o]; else h.hpre[o] = acc[t] + h.b1[o];
  }
}
__global__ __launch_bounds__(64) void k_value_out2(HeadT h) {   // one wave per batch row
  const int b = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int j = lane; j < h.FC; j += 64) { const float hv = h.hpre[(size_t)b * h.FC + j]; s += (hv > 0.f ? hv : 0.f) * h.W2[j]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) h.o[b] = s + h.b2[b];
}
// cost sums -> hacc[6] (policy), hacc[7] (value), zeroed by the caller; k_cost_fin writes cost[0..1]
__global__ __launch_bounds__(256) void k_cost_part(HeadT h, double* __restrict__ hacc) {
  __shared__ double red[2][4];
  double sp = 0, sv = 0;
  const int n = h.B * h.A;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) sp += -(double)(h.Pi[i] * h.logits[i] + (1.f - h.Pi[i]) * (1.f - h.logits[i]));
  for (int b = blockIdx.x * 256 + threadIdx.x; b < h.B; b += gridDim.x * 256) { const double d = h.o[b] - h.V[b]; sv += d * d; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sp += __shfl_xor(sp, o, 64); sv += __shfl_xor(sv, o, 64); }
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sp; red[1][threadIdx.x >> 6] = sv; }
  __syncthreads();
  if (threadIdx.x < 2) atomicAdd(&hacc[6 + threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}
__global__ void k_cost_fin(HeadT h, const double* __restrict__ hacc) {
  if (threadIdx.x == 0) { h.cost[0] = (float)(hacc[6] / ((double)h.B * h.A)); h.cost[1] = (float)(hacc[7] / h.B); }
}
// dW[i][j] = sum_b y[b][i] * d[b][j], eight rows i per block: grid (ceil(N / 256), ceil(K / 8), 2); z = 0: dWp (d = the xent gradient),
// z = 1: dW1 (d = the value head's hidden-layer gradient).  The b loop is the old kernel's.
__global__ __launch_bounds__(256) void k_fc_bwd_w(HeadT h) {
  extern __shared__ float ys[];                      // [B][8]
  const bool pol = blockIdx.z == 0;
  const int N = pol ? h.A : h.FC, K = pol ? 2 * h.HW : h.HW;
  const int j = blockIdx.x * 256 + threadIdx.x, i0 = blockIdx.y * 8;
  if (blockIdx.x * 256 >= N || i0 >= K) return;
  for (int e = threadIdx.x; e < h.B * 8; e += 256) {
    const int b = e >> 3, t = e & 7, i = i0 + t;
    ys[e] = i < K ? h.yh[((size_t)b * 3 + (pol ? 0 : 2)) * h.HW + i] : 0.f;
  }
  __syncthreads();
  if (j >= N) return;
  const float sc = 1.0f / ((float)h.B * (float)h.A);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int b = 0; b < h.B; b++) {
    float d;
    if (pol) d = (1.f - 2.f * h.Pi[(size_t)b * h.A + j]) * sc;
    else { const float dob = 2.f * (h.o[b] - h.V[b]) / (float)h.B; d = h.hpre[(size_t)b * h.FC + j] > 0.f ? dob * h.W2[j] : 0.f; }
#pragma unroll
    for (int t = 0; t < 8; t++) acc[t] += ys[b * 8 + t] * d;
  }
  float* dW = pol ? h.dWp : h.dW1;
#pragma unroll
  for (int t = 0; t < 8; t++) if (i0 + t < K) dW[(size_t)(i0 + t) * N + j] = acc[t];
}
// the elementwise / small parts of the FC backward: dbp, db1, dW2, db2
__global__ void k_fc_bwd_small(HeadT h) {
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const float sc = 1.0f / ((float)h.B * (float)h.A);
  const int n_bp = h.B * h.A, n_b1 = h.B * h.FC, n_w2 = h.FC, n_b2 = h.B;
  if (idx < n_bp) { h.dbp[idx] = (1.f - 2.f * h.Pi[idx]) * sc; return; }
  idx -= n_bp;
  if (idx < n_b1) {
    const int b = idx / h.FC, j = idx - b * h.FC;
    const float dob = 2.f * (h.o[b] - h.V[b]) / (float)h.B;
    h.db1[idx] = h.hpre[idx] > 0.f ? dob * h.W2[j] : 0.f;
    return;
  }
  idx -= n_b1;
  if (idx < n_w2) {
    float s = 0.f;
    for (int b = 0; b < h.B; b++) { const float hv = h.hpre[(size_t)b * h.FC + idx]; s += (2.f * (h.o[b] - h.V[b]) / (float)h.B) * (hv > 0.f ? hv : 0.f); }
    h.dW2[idx] = s;
    return;
  }
  idx -= n_w2;
  if (idx < n_b2) h.db2[idx] = 2.f * (h.o[idx] - h.V[idx]) / (float)h.B;
}
// d(yh)[b][q] for eight batch rows per block: grid (ceil(Q / 256), ceil(B / 8), 2); z = 0: q < 2 HW (sum over the A logits), z = 1: the value
// channel (sum over the FC hidden units).  The j loop is the old kernel's; the per-(b, j) factor comes from an LDS tile.
__global__ __launch_bounds__(256) void k_fc_bwd_y(HeadT h) {
  extern __shared__ float ds[];                      // [8][J]
  const bool pol = blockIdx.z == 0;
  const int J = pol ? h.A : h.FC, Q = pol ? 2 * h.HW : h.HW;
  const int q = blockIdx.x * 256 + threadIdx.x, b0 = blockIdx.y * 8;
  if (blockIdx.x * 256 >= Q) return;
  const float sc = 1.0f / ((float)h.B * (float)h.A);
  for (int e = threadIdx.x; e < 8 * J; e += 256) {
    const int t = e / J, j = e - t * J, b = b0 + t;
    float d = 0.f;
    if (b < h.B) {
      if (pol) d = (1.f - 2.f * h.Pi[(size_t)b * h.A + j]) * sc;
      else { const float dob = 2.f * (h.o[b] - h.V[b]) / (float)h.B; d = h.hpre[(size_t)b * h.FC + j] > 0.f ? dob * h.W2[j] : 0.f; }
    }
    ds[e] = d;
  }
  __syncthreads();
  if (q >= Q) return;
  const float* Wr = (pol ? h.Wp : h.W1) + (size_t)q * J;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int j = 0; j < J; j++) {
    const float w = Wr[j];
#pragma unroll
    for (int t = 0; t < 8; t++) acc[t] += ds[t * J + j] * w;
  }
#pragma unroll
  for (int t = 0; t < 8; t++) {
    const int b = b0 + t;
    if (b < h.B) h.dyh[(size_t)b * 3 * h.HW + (pol ? 0 : 2 * h.HW) + q] = acc[t];
  }
}
// head BatchNorm backward in two passes over all rows: (1) dgamma / dbeta, dxh -> dzh, the two channel sums -> hacc[8..13] (zeroed by the
// caller); (2) dzh = inv * (dxh - s1 / m - xh * s2 / m)
__global__ __launch_bounds__(256) void k_head_bn_bwd_a(TGeo g, const float* __restrict__ zh, const float* __restrict__ yh, const float* __restrict__ dyh,
                                                       const float* __restrict__ hg, const float* __restrict__ mean, const float* __restrict__ inv,
                                                       float* __restrict__ dhg, float* __restrict__ dhb, float* __restrict__ dzh, double* __restrict__ hacc) {
  __shared__ double red[6][4];
  double a1[3] = {0, 0, 0}, a2[3] = {0, 0, 0};
  for (int r = blockIdx.x * 256 + threadIdx.x; r < g.M; r += gridDim.x * 256) {
    const int b = r / g.HW, p = r - b * g.HW;
#pragma unroll
    for (int j = 0; j < 3; j++) {
      const size_t o = ((size_t)b * 3 + j) * g.HW + p;
      const float gg = yh[o] > 0.f ? dyh[o] : 0.f;
      const float xh = (zh[(size_t)r * 3 + j] - mean[j]) * inv[j];
      dhg[o] = gg * xh; dhb[o] = gg;
      const float dxh = gg * hg[o];
      dzh[(size_t)r * 3 + j] = dxh;
      a1[j] += dxh; a2[j] += (double)dxh * xh;
    }
  }
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a1[j] += __shfl_xor(a1[j], o, 64); a2[j] += __shfl_xor(a2[j], o, 64); }
  if ((threadIdx.x & 63) == 0) { for (int j = 0; j < 3; j++) { red[j][threadIdx.x >> 6] = a1[j]; red[3 + j][threadIdx.x >> 6] = a2[j]; } }
  __syncthreads();
  if (threadIdx.x < 6) atomicAdd(&hacc[8 + threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}
__global__ __launch_bounds__(256) void k_head_bn_bwd_b(TGeo g, const float* __restrict__ zh, const float* __restrict__ mean, const float* __restrict__ inv,
                                                       float* __restrict__ dzh, const double* __restrict__ hacc) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)g.M * 3) return;
  const int j = (int)(idx % 3);
  const float s1 = (float)hacc[8 + j], s2 = (float)hacc[11 + j], m = (float)g.M;
  const float xh = (zh[idx] - mean[j]) * inv[j];
  dzh[idx] = inv[j] * (dzh[idx] - s1 / m - xh * (s2 / m));
}
// dhc[j][c] += sum over the block's rows: one thread per channel, the three j at once (x read once), four rows in flight
__global__ __launch_bounds__(256) void k_head_conv_bwd_w2(TGeo g, const float* __restrict__ x, const float* __restrict__ dzh, float* __restrict__ dhc, int Kp,
                                                          int rows_per_block) {
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, g.M);
  for (int c = threadIdx.x; c < Kp; c += 256) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int r = r0; r < r1; r++) {
      const float xv = x[pix_off(g, r) * Kp + c];
      s0 += dzh[(size_t)r * 3] * xv; s1 += dzh[(size_t)r * 3 + 1] * xv; s2 += dzh[(size_t)r * 3 + 2] * xv;
    }
    atomicAdd(&dhc[c], s0); atomicAdd(&dhc[Kp + c], s1); atomicAdd(&dhc[2 * Kp + c], s2);
  }
}
__global__ __launch_bounds__(256) void k_head_conv_bwd_x2(TGeo g, const float* __restrict__ dzh, const float* __restrict__ hc, float* __restrict__ dx, int Kp) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;   // one float4 of one row
  const int k4 = Kp >> 2;
  if (idx >= (size_t)g.M * k4) return;
  const int r = (int)(idx / k4), c = (int)(idx - (size_t)r * k4) * 4;
  const float d0 = dzh[(size_t)r * 3], d1 = dzh[(size_t)r * 3 + 1], d2 = dzh[(size_t)r * 3 + 2];
  const float4 a = *reinterpret_cast<const float4*>(hc + c), b = *reinterpret_cast<const float4*>(hc + Kp + c), e = *reinterpret_cast<const float4*>(hc + 2 * Kp + c);
  float4 o;
  o.x = d0 * a.x + d1 * b.x + d2 * e.x; o.y = d0 * a.y + d1 * b.y + d2 * e.y; o.z = d0 * a.z + d1 * b.z + d2 * e.z; o.w = d0 * a.w + d1 * b.w + d2 * e.w;
  *reinterpret_cast<float4*>(dx + pix_off(g, r) * Kp + c) = o;
}

}  // namespace agz

using namespace agz;

// ---- forward convolution of a dual block with BOTH operands DMA'd into LDS (AGZ_COMPUTE_WINO_H2, round 5) -------------------------------
// conv_h2.hpp's conv3x3_h2w_kernel splits its fp32 activations while staging them: every element is scaled, split and written to LDS once
// per tap and column tile (18 times per layer), and that VALU + ds_write issue — not operand bytes — bounds it (0.67 ms per G19 layer; the
// same kernel on 256-row tiles, with half the weight re-reads, ran 0.665).  Measured here: 0.583 ms per G19 layer.  The layer input is
// split ONCE into hi / lo fp16 planes
// (k_split_h2p, the tensor's own power-of-two range from the BatchNorm pass that wrote it) — the very planes the weight gradient of the
// same layer reads in the backward pass, which therefore no longer splits x itself — and the convolution is a pure DMA GEMM like
// wino_gemm_h2g_kernel: 128 x 256 tile, K step = 32 channels of one tap, `buffer_load ... lds` of 16 bytes per lane for both operands
// (A rows gathered by padded pixel: one lane offset per 16-row instruction, the tap and the channel chunk in the scalar offset), single
// 48 KB stage, plain __syncthreads(), three workgroups per CU (two, without the 3 spilled registers: 0.596 against 0.583 ms).  LDS image:
// four pieces (A hi, A lo, B hi, B lo) of 64-byte rows, 16-byte
// units XOR-swizzled with (row >> 2) & 3 — on the SOURCE address, the DMA writes lane-linearly.  Products as everywhere: lo*hi, hi*lo, hi*hi.
// (conv_h2.hpp's scale rule for the weight image: s = 2^(13 - floor(log2(amax))), exact inverse)
__device__ __forceinline__ void h2_scales(unsigned amax_bits, float* s, float* inv) {
  int e = (int)((amax_bits >> 23) & 0xffu);
  if (amax_bits == 0u) { *s = 1.f; *inv = 1.f; return; }
  e = e < 30 ? 30 : (e > 230 ? 230 : e);
  *s = __uint_as_float((unsigned)(267 - e) << 23);
  *inv = __uint_as_float((unsigned)(e - 13) << 23);
}
struct ConvDmaArgs {
  const _Float16 *xh, *xl;   // [B][Hp][Wp][Cin] hi / lo planes of the layer input (zero halo)
  const _Float16* w2;        // [Cin/32][tap][piece][Ntot][32]
  float* y;                  // [B][Hp][Wp][Ntot] raw GEMM result (halo untouched)
  const unsigned* x_amax;    // bits of max|x| (the planes' scale: wg_h2_scale)
  const unsigned* w_amax;    // bits of max|w| (the weight image's scale: h2_scales of conv_h2.hpp)
  TGeo g;
  int Cin, Ntot, n_mtiles, n_ntiles;
};
__device__ __forceinline__ unsigned cd_lds_off(int row, int unit) { return cmaps::lds_off(row, unit); }   // (conv_maps.hpp: shared with the CPU check)
__global__ __launch_bounds__(256, 3) void k_conv_h2dma(ConvDmaArgs a) {
  constexpr int PA = 128 * 64, PB = 256 * 64;          // bytes of one A / B piece image
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * PA + 2 * PB];   // 48 KB

  const int nblk = a.n_mtiles * a.n_ntiles;
  const int id = blockIdx.x;
  int q = nblk >> 3, rr = nblk & 7, xcd = id & 7, slot = id >> 3;
  int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + slot;
  const int m_tile = tile / a.n_ntiles, n_tile = tile - m_tile * a.n_ntiles;
  const int m0 = m_tile * 128, n0 = n_tile * 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;

  const size_t plane_bytes = (size_t)a.g.B * a.g.Hp * a.g.Wp * a.Cin * 2;
  const __amdgpu_buffer_rsrc_t rxh = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.xh), 0, (int)plane_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rxl = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.xl), 0, (int)plane_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.w2), 0, 0x7fffffff, 0x00020000);

  // DMA lane mapping: an instruction lands 16 rows x 64 bytes; lane l -> row 16 j + (l >> 2), LDS unit l & 3, source unit (l & 3) ^ ((l >> 4) & 3)
  const unsigned src_unit = (unsigned)((lane & 3) ^ ((lane >> 4) & 3)) << 4;
  unsigned voa[2];                                       // A: rows 32 wid + 16 j + (lane >> 2) -> padded pixel, shifted by -(Wp + 1) (the taps add 0 .. 2 Wp + 2)
#pragma unroll
  for (int j = 0; j < 2; j++) {
    int m = m0 + 32 * wid + 16 * j + (lane >> 2);
    if (m >= a.g.M) m = a.g.M - 1;
    voa[j] = (unsigned)((pix_off(a.g, m) - (size_t)(a.g.Wp + 1)) * (size_t)a.Cin * 2) + src_unit;
  }
  int nrow = n0 + 64 * wid + (lane >> 2);                // B: rows 64 wid + 16 j + (lane >> 2): + j KB in the scalar offset
  const unsigned vob = (unsigned)(nrow < a.Ntot ? nrow : a.Ntot - 1) * 64u + src_unit;
  const unsigned piece_bytes = (unsigned)a.Ntot * 64u;
  unsigned char* const la = lds + wid * 2048;            // this wave's 32 rows of an A piece
  unsigned char* const lb = lds + 2 * PA + wid * 4096;   // this wave's 64 rows of a B piece

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
  // columns of this wave inside the 256-column tile: j = 0..3 -> 32-column groups of its 128 columns (the weight image's own order)
#pragma unroll
  for (int j = 0; j < 4; j++) rb[j] = wn * 128 + j * 32 + (lane & 31);
  const int kh = lane >> 5;

  const int NC = a.Cin >> 5;
  unsigned wso = 0;                                      // weight image: K steps in [chunk][tap] order, 2 pieces each
#pragma nounroll
  for (int cc = 0; cc < NC; cc++) {
#pragma nounroll
    for (int tap = 0; tap < 9; tap++) {
      const int ky = tap / 3, kx = tap - 3 * ky;
      const unsigned aso = (unsigned)((ky * a.g.Wp + kx) * a.Cin * 2 + cc * 64);
#pragma unroll
      for (int j = 0; j < 2; j++) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rxh, (wg_lds_ptr_t)(la + j * 1024), 16, voa[j], aso, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rxl, (wg_lds_ptr_t)(la + PA + j * 1024), 16, voa[j], aso, 0, 0);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (wg_lds_ptr_t)(lb + j * 1024), 16, vob, wso + (unsigned)j * 1024u, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (wg_lds_ptr_t)(lb + PB + j * 1024), 16, vob, wso + piece_bytes + (unsigned)j * 1024u, 0, 0);
      }
      wso += 2u * piece_bytes;
      __syncthreads();                                   // (its fence waits vmcnt(0): this wave's DMA has landed; then every wave's)
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        wg_f16x8_t A_[2][2];
#pragma unroll
        for (int p = 0; p < 2; p++)
#pragma unroll
          for (int i = 0; i < 2; i++) A_[i][p] = *reinterpret_cast<const wg_f16x8_t*>(lds + p * PA + cd_lds_off(ra[i], 2 * ks + kh));
#pragma unroll
        for (int j = 0; j < 4; j++) {                    // B fragments just in time
          const wg_f16x8_t b0 = *reinterpret_cast<const wg_f16x8_t*>(lds + 2 * PA + cd_lds_off(rb[j], 2 * ks + kh));
          const wg_f16x8_t b1 = *reinterpret_cast<const wg_f16x8_t*>(lds + 2 * PA + PB + cd_lds_off(rb[j], 2 * ks + kh));
#pragma unroll
          for (int i = 0; i < 2; i++) {                  // small terms first: lo*hi, hi*lo, hi*hi
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[i][1], b0, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[i][0], b1, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[i][0], b0, acc[i][j], 0, 0, 0);
          }
        }
      }
      __syncthreads();                                   // every wave has read the stage before the next DMA overwrites it
    }
  }
  // raw result, un-scaled by the two exact powers of two
  float sw_, unw_;
  h2_scales(*a.w_amax, &sw_, &unw_);
  const float un = unw_ / wg_h2_scale(*a.x_amax);
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = m0 + (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m >= a.g.M) continue;
      float* yr = a.y + pix_off(a.g, m) * (size_t)a.Ntot;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const int c = n0 + rb[j];
        if (c < a.Ntot) yr[c] = acc[i][j][r] * un;
      }
    }
}

// ---- the same convolution, nine taps from ONE x image (round 5, second form) ---------------------------------------------------------------
// k_conv_h2dma above streams 48 KB through L2 -> LDS per K step of 6.3 MFLOP: 5 GB per G19 layer, which at the ~10 TB/s the L2 delivers to
// the CUs is the whole 0.56 ms (46 % of the fp16x2 MFMA roof).  Here a workgroup computes 256 output pixels x 128 output channels and
// lands, per 32-channel chunk, ONE image of its input: the padded pixels from (first output pixel) - Wp - 1 to (last output pixel) + Wp + 1,
// hi and lo plane — tap (ky, kx) of output pixel m reads image row ra(m) + ky Wp + kx, ra(m) = padded distance of m from the tile's first
// pixel (row ends and one board crossing included: <= 372 rows for 19x19) — 47 KB for nine taps; the weights stream per tap (16 KB,
// double-buffered, landing under the tap before).  21 KB per K step instead of 48.  Same products in the same order per accumulator
// (chunks ascending, taps ascending, lo*hi, hi*lo, hi*hi): results are k_conv_h2dma's bit for bit.  80 KB of LDS: two workgroups per CU;
// assembly DMA + counted waits as in k_wgrad_h2t3 (the compiler's own LDS-DMA bookkeeping would drain the weight prefetch before every tap).
// Measured: G19 step 43.1 -> 42.2 ms (forward_backward, same box; 0.565 -> ~0.52 ms per layer), far from the 2.3x fewer bytes.  Decomposition
// (MODE, scripts/train_bench.py --fb --hooks 1,65,129,193; ms per step): everything 42.2, no x image after chunk 0 41.9, no weight DMA
// 37.5, neither 36.9 — the arithmetic alone runs AT the MFMA rate (0.255 ms per layer), the image costs 17 us per layer, the weights' 16 KB
// per tap 235 us.  That cost did not move with: no look-ahead at all (+0.2 ms per step); no wait for the weights (-2.3 of the 4.7 ms, wrong
// results); an XCD per n-tile (its quarter of the weights resident in its L2); every K step landing the SAME lines; a different chunk
// order per m-tile; nor with the weights loaded straight into registers one tap ahead (lane = output channel, no LDS, no barrier inside a
// chunk: 42.1 ms, and 45.8 on every other run).  Vector-memory traffic beside 48 MFMAs + 24 fragment reads per tap slows the tap whatever
// its source, latency and path; what remains is fewer weight bytes per product (a 256 x 256 tile: one workgroup per CU, another kernel).
constexpr int CD3_IMG = cmaps::CD3_IMG;
template <int MODE>   // (decomposition runs only — WRONG results: bit 0 = the x image lands for chunk 0 only, bit 1 = no weight DMA after K step 0)
__global__ __launch_bounds__(256, 2) void k_conv_h2dma3(ConvDmaArgs a) {
  constexpr int PA = CD3_IMG * 64;          // bytes of one piece (hi or lo) of the x image
  constexpr int PB = 128 * 64, SB = 2 * PB;  // one piece of a tap's weights; a weight stage (hi, lo)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * PA + 2 * SB];   // 47 616 + 32 768 bytes

  const int nblk = a.n_mtiles * a.n_ntiles;
  const int id = blockIdx.x;
  int q = nblk >> 3, rr = nblk & 7, xcd = id & 7, slot = id >> 3;
  int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + slot;
  const int m_tile = tile / a.n_ntiles, n_tile = tile - m_tile * a.n_ntiles;   // (the n-tiles of an m-tile are neighbours on one XCD: one x image in its L2)
  const int m0 = m_tile * 256, n0 = n_tile * 128;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;

  const size_t plane_bytes = (size_t)a.g.B * a.g.Hp * a.g.Wp * a.Cin * 2;
  auto mk = [](const void* p, size_t bytes) -> wg_rsrc_t {
    const unsigned long long u = (unsigned long long)p;
    wg_rsrc_t r;
    r[0] = __builtin_amdgcn_readfirstlane((unsigned)u); r[1] = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane((unsigned)bytes); r[3] = 0x00020000u;
    return r;
  };
  const wg_rsrc_t rxh = mk(a.xh, plane_bytes), rxl = mk(a.xl, plane_bytes);
  const wg_rsrc_t rw = mk(a.w2, (size_t)(a.Cin >> 5) * 9 * 2 * a.Ntot * 64);
  const unsigned lds0 = (unsigned)(size_t)(wg_lds_ptr_t)lds;

  // DMA lane mapping (both operands): an instruction lands 16 rows x 64 bytes lane-linearly; lane l -> row 16 j + (l >> 2), LDS unit l & 3,
  // SOURCE unit (l & 3) ^ ((l >> 4) & 3) (the read side's swizzle, cd_lds_off)
  const unsigned src_unit = cmaps::dma_src_unit(lane);
  const size_t pix0 = pix_off(a.g, m0);
  const int pb = cmaps::cd3_base(pix0, a.g.Wp);                             // padded pixel of image row 0 (>= 0)
  const unsigned va0 = (unsigned)(pb + cmaps::dma_row(lane, 0)) * (unsigned)(a.Cin * 2) + src_unit;
  const unsigned vb0 = (unsigned)(n0 + 32 * wid + cmaps::dma_row(lane, 0)) * 64u + src_unit;
  const unsigned piece_bytes = (unsigned)a.Ntot * 64u;
  auto issue_A = [&](int cc) {                       // instructions j = wid, wid + 4, .. < 24; rows >= CD3_IMG stay unwritten (lanes off)
#pragma unroll
    for (int jj = 0; jj < 6; jj++) {
      const int j = wid + 4 * jj;
      if (cmaps::dma_row(lane, j) < CD3_IMG) {
        const unsigned vo = va0 + (unsigned)j * (unsigned)(16 * a.Cin * 2) + (unsigned)cc * 64u;
        wg_dma16(rxh, lds0 + cmaps::dma_dst(0, j), vo);
        wg_dma16(rxl, lds0 + (unsigned)PA + cmaps::dma_dst(0, j), vo);
      }
    }
  };
  auto issue_B = [&](int kstep, int buf) {           // K step = chunk * 9 + tap: the weight image's own order, two pieces each
    const unsigned so = (unsigned)kstep * 2u * piece_bytes;
    const unsigned d = lds0 + 2u * PA + (unsigned)buf * SB + (unsigned)wid * 2048u;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      wg_dma16(rw, d + cmaps::dma_dst(0, j), vb0 + so + (unsigned)j * 1024u);
      wg_dma16(rw, d + PB + cmaps::dma_dst(0, j), vb0 + so + piece_bytes + (unsigned)j * 1024u);
    }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  int ra[4];                                           // image row of this lane's output pixel under tap (0, 0)
#pragma unroll
  for (int i = 0; i < 4; i++) {
    int m = m0 + (wm * 4 + i) * 32 + (lane & 31);
    if (m >= a.g.M) m = a.g.M - 1;
    ra[i] = cmaps::cd3_row0(pix_off(a.g, m), pix0);
  }
  const int kh = lane >> 5;
  unsigned ob[2][2];                                   // weight fragment offsets inside a stage piece
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int ks = 0; ks < 2; ks++) ob[j][ks] = cd_lds_off(wn * 64 + j * 32 + (lane & 31), 2 * ks + kh);

  const int NC = a.Cin >> 5, NS = NC * 9;
  issue_B(0, 0);
  int t = 0;
#pragma nounroll
  for (int cc = 0; cc < NC; cc++) {
    if (cc > 0) __builtin_amdgcn_s_barrier();          // every wave is done with the image of chunk cc - 1
    if (!(MODE & 1) || cc == 0) issue_A(cc);
#pragma nounroll
    for (int tap = 0; tap < 9; tap++) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the tap's weights (and, at tap 0, of the image) has landed
      __builtin_amdgcn_s_barrier();                      // ... every wave's; and every wave is done with the tap before
      __builtin_amdgcn_sched_barrier(0);
      if (!(MODE & 2) && t + 1 < NS) issue_B(t + 1, (t + 1) & 1);
      const unsigned char* bs = lds + 2 * PA + (t & 1) * SB;
      const int ky = (tap * 11) >> 5;                    // tap / 3 for tap < 9
      const int toff = cmaps::cd3_tap(ky, tap - 3 * ky, a.g.Wp);
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        wg_f16x8_t B_[2][2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
          B_[j][0] = *reinterpret_cast<const wg_f16x8_t*>(bs + ob[j][ks]);
          B_[j][1] = *reinterpret_cast<const wg_f16x8_t*>(bs + PB + ob[j][ks]);
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int row = ra[i] + toff;
          const unsigned o = cmaps::lds_off(row, 2 * ks + kh);
          const wg_f16x8_t ah = *reinterpret_cast<const wg_f16x8_t*>(lds + o);
          const wg_f16x8_t al = *reinterpret_cast<const wg_f16x8_t*>(lds + PA + o);
#pragma unroll
          for (int j = 0; j < 2; j++) {                  // small terms first: lo*hi, hi*lo, hi*hi
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, B_[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, B_[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, B_[j][0], acc[i][j], 0, 0, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      t++;
    }
  }
  // raw result, un-scaled by the two exact powers of two
  float sw_, unw_;
  h2_scales(*a.w_amax, &sw_, &unw_);
  const float un = unw_ / wg_h2_scale(*a.x_amax);
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int m = m0 + (wm * 4 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (m >= a.g.M) continue;
      float* yr = a.y + pix_off(a.g, m) * (size_t)a.Ntot + n0 + wn * 64 + (lane & 31);
#pragma unroll
      for (int j = 0; j < 2; j++) yr[j * 32] = acc[i][j][r] * un;
    }
}

struct TLayer {
  int Cin_p, Cout_p, nbr;
  size_t o_wf, o_gamma, o_beta;   // offsets in the flat P / G buffers
  float *wt = nullptr, *z = nullptr, *out = nullptr, *mean = nullptr, *inv = nullptr;
  unsigned short *w3f = nullptr, *w3t = nullptr;   // bf16x3 images of the forward / transposed filter (AGZ_COMPUTE_BF16X3)
  _Float16* xh2 = nullptr;                         // AGZ_COMPUTE_WINO_H2: hi plane then lo plane of the layer INPUT (forward DMA convolution + weight gradient)
  bool x_planes = false;                           // ... and they hold this step's input (set by the forward pass, read by the backward pass)
  // weight images of the step, built on the side stream beside the forward pass (prep_weights): fw = fp16x2 image of the filter for the DMA
  // forward convolution, bw = Winograd image of the transposed filter for the data gradient
  agz::RawWeights fw, bw;
  hipEvent_t ev_fw = nullptr;
  bool fw_ready = false, bw_ready = false;         // ... built for THIS step
};
struct TParamRef { std::string name; int kind; std::vector<int> shape; int layer; int sub; };  // sub: 0 filter(a) 1 gamma 2 beta, for dual +10 = branch b

struct agz_trainer {
  agz_ctx* ctx = nullptr;
  agz_net_conf conf{};
  TGeo g{};
  int K, Kp, Fp = 32, L, A, FC, B, F;
  std::vector<TLayer> layers;          // 0 = init, 1..L = blocks
  std::vector<TParamRef> prefs;        // reference Model() order
  size_t n_flat = 0;
  float *P = nullptr, *G = nullptr;    // flat learnables / gradients
  // head offsets in flat
  size_t o_hc, o_hg, o_hb, o_Wp, o_bp, o_W1, o_b1, o_W2, o_b2;
  // work buffers
  float *x0 = nullptr, *dA = nullptr, *dB = nullptr, *dz = nullptr, *dz0 = nullptr;
  double* acc = nullptr;               // [2][1024] channel sums
  float *zh = nullptr, *yh = nullptr, *dyh = nullptr, *dzh = nullptr, *hmean = nullptr, *hinv = nullptr;
  float *logits = nullptr, *hpre = nullptr, *o = nullptr, *cost = nullptr;
  float *d_planes = nullptr, *d_pi = nullptr, *d_v = nullptr;
  bool x3 = false, x3_force = false;   // agz_trainer_set_compute_mode (FORCE: also below the chip-filling threshold, tests)
  // AGZ_COMPUTE_WINO_H2: the DATA-GRADIENT convolutions of the dual blocks through conv_wino_h2.hpp; the forward convolutions are DIRECT
  // fp16x2 convolutions (use_h2_fwd / dma_layer below).  (Round 2 also ran the forward convolutions through the Winograd path: 72 instead
  // of 87 ms per G19 step, but its rounding — 2e-6 of the output rms against 2e-7 — flips a ReLU unit against the reference arithmetic on
  // every other data draw at K = 256 / 19x19, which moves that unit's gradients by percent: outside the stated tolerance, removed.)
  bool wino = false;
  WinoRawScratch wsc;
  // ... and the weight gradient with fp16x2 products: both operands split once per layer (k_wgrad_h2)
  unsigned *dz_h2 = nullptr, *x_h2 = nullptr;
  // per-layer scratch cleared ONCE per step (one memset each instead of one per layer): the backward BatchNorm channel sums
  // [L + 1][2][1024] and the fp16x2 weight gradient's range words [L + 2][2] = {max|dz| of layer l, max|x| of its input}
  double* acc_b = nullptr;
  unsigned* amax_words = nullptr;
  unsigned* board_words = nullptr;     // [2][L + 2][B] per-board ranges: [0] layer inputs (from k_bn_apply_v), [1] dz (from k_bn_bwd2_v)
  std::vector<char> x_amax_ready;      // layer l's input range was produced by the forward pass (k_bn_apply_v)
  std::vector<char> xb_ready;          // ... and its per-board ranges (board_words)
  size_t* zero_tab = nullptr;          // [2][L + 1] offsets and counts of the filter-gradient regions (k_zero_regions)
  // the weight gradient of a layer (matrix pipe / L2 bound, ~0.35 GB of HBM traffic) runs on its own stream beside the same layer's data
  // gradient and the next layer's BatchNorm backward (HBM bound): both only need dz.  Measured 50.6-50.9 -> 49.1-50.0 ms per G19 step —
  // the weight gradient's two workgroups per CU hold 496 of a SIMD's 512 registers, so little else becomes resident beside them (forcing
  // ONE workgroup per CU to make room: 49.4-50.4, no better)
  hipStream_t wg_stream = nullptr;
  hipEvent_t ev_dz = nullptr, ev_split = nullptr, ev_join = nullptr;
  size_t dz_h2_cap = 0, x_h2_cap = 0;
  std::vector<char> planes_fused;      // layer l's input planes were written by the previous layer's k_bn_apply_v this step
  bool dma_layer(int l) const {        // layer l's forward convolution runs as the DMA GEMM on pre-split planes (k_conv_h2dma)
    const TLayer& ly = layers[l];
    return dma_fwd && use_h2_fwd(ly.Cin_p, ly.Cout_p) && ly.Cin_p % 32 == 0 && ly.Cout_p % 256 == 0 && g.W >= 16 &&
           (size_t)B * g.Hp * g.Wp * ly.Cin_p * 2 < ((size_t)1 << 31);
  }
  bool use_wino(int cin, int cout) const {
    return wino && cin % 32 == 0 && cin >= 64 && conv3x3_raw_wino_h2_fits(B, g.H, g.W, cin, cout) &&
           (x3_force || (size_t)((g.M + 127) / 128) * ((cout + 127) / 128) >= (size_t)ctx->num_cus);
  }
  // AGZ_COMPUTE_WINO_H2 forward: the DIRECT convolution with fp16x2 products (conv_h2.hpp's kernel with a raw store) — no transform in
  // front of the products, so its rounding is that of fp32 accumulation (3e-7 of the output rms, the same as bf16x3), unlike the
  // Winograd forward that was removed
  bool use_h2_fwd(int cin, int cout) const {
    return wino && cin >= 64 && conv3x3_raw_h2_fits(B, g.H, g.W, cin, cout) &&
           (x3_force || (size_t)((g.M + 127) / 128) * (cout / 256) >= (size_t)ctx->num_cus);
  }
  // bf16x3 only where the 128-row tiles fill the chip (same rule as inference) and the filter has whole 16-channel chunks
  bool use_x3(const TLayer& ly, int cin, int cout) const {
    return x3 && ly.w3f && cin % 16 == 0 && cin >= 64 &&
           (x3_force || (size_t)((g.M + 127) / 128) * ((cout + 127) / 128) >= (size_t)ctx->num_cus);
  }
  std::vector<void*> allocs;
  template <typename T> int alloc(T** p, size_t n) {
    void* q = nullptr;
    AGZ_HIP_TRY(hipMalloc(&q, n * sizeof(T)));
    AGZ_HIP_TRY(hipMemsetAsync(q, 0, n * sizeof(T), ctx->stream));
    allocs.push_back(q); *p = (T*)q; return AGZ_OK;
  }
  int forward_backward_dev(const float* planes_dev, const float* pi_dev, const float* v_dev);
  // data-parallel step (comm.hip, agz_trainer_forward_backward_allreduce): called once per slice of the flat gradient buffer as soon
  // as the kernels that write it have been ENQUEUED — heads first, then layer L .. 0 (the order of the backward pass, the same on
  // every rank) — with the stream whose completion means "slice written"; the slices tile [0, n_flat) exactly
  std::function<int(size_t off, size_t n, hipStream_t ready)> on_slice;
  unsigned* amax_prev = nullptr;   // [(L + 2) * 2] last step's amax_words: the range estimates k_bn_apply_v writes the next layer's planes under
  double* head_acc = nullptr;   // [16] partial sums of the head kernels' second form (statistics, cost, BatchNorm backward)
  bool fast_heads = true;   // agz_trainer_set_dma_forward(t, on | 2 * heads): the second form of the head kernels (A/B hook)
  bool dma_fwd = true;      // AGZ_COMPUTE_WINO_H2 forward convolutions through k_conv_h2dma (agz_trainer_set_dma_forward, agz_debug.h: A/B hook)
  bool hoist_w = true;      // ... every layer's weight images at the start of the step on the side stream (bit 3 of the same hook: per layer, in line)
  bool conv3 = true;        // bit 5 of the same hook set: k_conv_h2dma (a K step = one tap) instead of k_conv_h2dma3 (nine taps from one x image)
  int conv3_mode = 0;       // bits 6, 7 of the hook: k_conv_h2dma3<MODE> (decomposition runs: WRONG results)
  int conv3_span = -1;      // rows of k_conv_h2dma3's largest x image for this geometry (<= CD3_IMG or the kernel is not used)
  bool conv3_layer(int l) {
    const TLayer& ly = layers[l];
    if (!conv3 || !dma_layer(l) || ly.Cout_p % 128 != 0) return false;
    if (conv3_span < 0) {
      int mx = 0;
      for (int m0 = 0; m0 < g.M; m0 += 256)
        mx = std::max(mx, cmaps::cd3_rows(cmaps::pix(m0, g.HW, g.W, g.Hp, g.Wp), cmaps::pix(std::min(m0 + 255, g.M - 1), g.HW, g.W, g.Hp, g.Wp), g.Wp));
      conv3_span = mx;
    }
    return conv3_span <= CD3_IMG;
  }
  // k_wgrad_h2t3 at ONE workgroup per CU (round 6): 16 KB of unused dynamic LDS on top of its 73.7 KB make a second workgroup not fit.  Two per CU
  // hold the whole register file (243 registers x 2 waves per SIMD) and 147 of 160 KB of LDS: nothing of the main stream's chain (the data
  // gradient, the next layer's BatchNorm backward) could be resident beside it and the side stream overlapped 4 %.  One per CU runs 0.72 instead
  // of 0.53 ms on its own and leaves half of every SIMD's registers: both chains now progress together (the layer's backward 1.28 -> 1.19 ms,
  // the step 42.4 -> 41.0 ms on the same box; profiles/r06/train_overlap.md).  Bit 8 of the hook: two per CU again (A/B).
  int wg_pad_lds = 16384;
  bool wg_low_prio = false; // bit 9: the side stream at the lowest priority (set before the first step)
  bool one_stream = false;  // bit 4 of the same hook: no side stream at all (diagnostic: a kernel table without overlap shows every kernel's own duration)
  hipEvent_t ev_w0 = nullptr, ev_bw = nullptr;
  int side_stream();
  int prep_weights();
  float fuse_lr = 0.f;      // != 0 during a fused step: k_bn_bwd1 updates gamma / beta in place, apply() skips them
  bool fused_done = false;  // the backward that just ran took the fused path
};

static inline int nblk(size_t n, int bs = 256) { return (int)((n + bs - 1) / bs); }

// clears n regions of one buffer in a single launch (grid: x = blocks per region, y = region)
__global__ __launch_bounds__(256) void k_zero_regions(float* __restrict__ base, const size_t* __restrict__ tab, int n) {
  const size_t off = tab[blockIdx.y], cnt = tab[n + blockIdx.y];
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (size_t)gridDim.x * 256) base[off + i] = 0.f;
}

int agz_trainer::side_stream() {
  if (wg_stream) return AGZ_OK;
  if (wg_low_prio) {
    int lo = 0, hi = 0;
    AGZ_HIP_TRY(hipDeviceGetStreamPriorityRange(&lo, &hi));   // (numerically larger = lower priority)
    AGZ_HIP_TRY(hipStreamCreateWithPriority(&wg_stream, hipStreamNonBlocking, lo));
  } else
    AGZ_HIP_TRY(hipStreamCreateWithFlags(&wg_stream, hipStreamNonBlocking));
  AGZ_HIP_TRY(hipEventCreateWithFlags(&ev_dz, hipEventDisableTiming));
  AGZ_HIP_TRY(hipEventCreateWithFlags(&ev_split, hipEventDisableTiming));
  AGZ_HIP_TRY(hipEventCreateWithFlags(&ev_join, hipEventDisableTiming));
  AGZ_HIP_TRY(hipEventCreateWithFlags(&ev_w0, hipEventDisableTiming));
  AGZ_HIP_TRY(hipEventCreateWithFlags(&ev_bw, hipEventDisableTiming));
  return AGZ_OK;
}

// AGZ_COMPUTE_WINO_H2: the filters are constant inside a step (apply() runs after the backward pass), so the per-layer weight passes — range
// word + fp16 hi / lo image for the DMA forward convolution (2 kernels), transpose + Winograd image for the data gradient (3 kernels), each a
// few MB of work that cannot fill the chip: 17 + 73 us per layer in line at G19 — are queued here on the side stream, which is otherwise
// idle until the backward pass: forward images first, in layer order with an event each, then the data gradient's with one event
int agz_trainer::prep_weights() {
  for (auto& ly : layers) { ly.fw_ready = false; ly.bw_ready = false; }
  if (!wino || !hoist_w || one_stream) return AGZ_OK;
  int r = side_stream();
  if (r != AGZ_OK) return r;
  hipStream_t s = ctx->stream, sw = wg_stream;
  AGZ_HIP_TRY(hipEventRecord(ev_w0, s));              // (the previous step's apply() and its readers of the images)
  AGZ_HIP_TRY(hipStreamWaitEvent(sw, ev_w0, 0));
  for (int l = 0; l <= L; l++) {
    TLayer& ly = layers[l];
    if (!dma_layer(l)) continue;
    if (!ly.ev_fw) AGZ_HIP_TRY(hipEventCreateWithFlags(&ly.ev_fw, hipEventDisableTiming));
    if ((r = agz::conv3x3_raw_h2_weights_to(ctx, sw, P + ly.o_wf, ly.Cin_p, ly.Cout_p, &ly.fw)) != AGZ_OK) return r;
    AGZ_HIP_TRY(hipEventRecord(ly.ev_fw, sw));
    ly.fw_ready = true;
  }
  for (int l = L; l >= 1; l--) {
    TLayer& ly = layers[l];
    const int C = ly.Cout_p;
    if (!use_wino(C, ly.Cin_p)) continue;
    hipLaunchKernelGGL(k_make_wt, dim3((unsigned)(((size_t)9 * C * ly.Cin_p + 255) / 256)), dim3(256), 0, sw, P + ly.o_wf, ly.wt, C, ly.Cin_p);
    if ((r = agz::conv3x3_raw_wino_h2_weights_to(ctx, sw, ly.wt, g.H, g.W, C, ly.Cin_p, &ly.bw)) != AGZ_OK) return r;
    ly.bw_ready = true;
  }
  AGZ_HIP_TRY(hipEventRecord(ev_bw, sw));
  return AGZ_OK;
}

int agz_trainer::forward_backward_dev(const float* planes, const float* pi, const float* v) {
  hipStream_t s = ctx->stream;
  const int RPB = 64;
  { int r0 = prep_weights(); if (r0 != AGZ_OK) return r0; }
  hipLaunchKernelGGL(k_pack_planes_t, dim3(nblk((size_t)g.M * Fp)), dim3(256), 0, s, planes, x0, g, F, Fp);
  // zero the gradient regions that are ACCUMULATED into (filters: atomics; heads).  The batch-shaped gamma / beta gradients — 98 % of
  // the flat buffer — are plain stores of every element (k_bn_bwd1) and need no clearing (one 7.7 GB memset per G19 step saved)
  hipLaunchKernelGGL(k_zero_regions, dim3(64, L + 1), dim3(256), 0, s, G, zero_tab, L + 1);
  AGZ_HIP_TRY(hipMemsetAsync(G + o_hc, 0, (n_flat - o_hc) * sizeof(float), s));
  AGZ_HIP_TRY(hipMemsetAsync(acc_b, 0, (size_t)(L + 1) * 2048 * sizeof(double), s));
  AGZ_HIP_TRY(hipMemcpyAsync(amax_prev, amax_words, (size_t)(L + 2) * 2 * sizeof(unsigned), hipMemcpyDeviceToDevice, s));   // (last step's ranges: estimates for the fused plane split)
  AGZ_HIP_TRY(hipMemsetAsync(amax_words, 0, (size_t)(L + 2) * 2 * sizeof(unsigned), s));
  AGZ_HIP_TRY(hipMemsetAsync(board_words, 0, (size_t)2 * (L + 2) * B * sizeof(unsigned), s));
  // ---- forward, training-mode BN
  const float* cur = x0;
  for (int l = 0; l <= L; l++) {
    TLayer& ly = layers[l];
    int r;
    ly.x_planes = false;
    const size_t n_x_fwd = (size_t)B * g.Hp * g.Wp * ly.Cin_p;
    if (dma_layer(l) && x_amax_ready[l]) {
      // planes of the layer input (its range word came out of the previous layer's BatchNorm pass, and so did the planes themselves
      // unless the range's exponent moved since the last step), then the DMA GEMM
      if (!ly.xh2) { r = alloc(&ly.xh2, 2 * n_x_fwd); if (r != AGZ_OK) return r; planes_fused[l] = 0; }
      const unsigned gs = (unsigned)std::min<size_t>(nblk(n_x_fwd / 4), (size_t)ctx->num_cus * 8);
      hipLaunchKernelGGL(k_split_h2p_cond, dim3(gs), dim3(256), 0, s, cur, ly.xh2, ly.xh2 + n_x_fwd, n_x_fwd / 4, amax_words + 2 * l + 1,
                         planes_fused[l] ? (const unsigned*)(amax_prev + 2 * l + 1) : (const unsigned*)nullptr);
      const void* w2 = nullptr; const unsigned* wmax = nullptr;
      if (ly.fw_ready) { AGZ_HIP_TRY(hipStreamWaitEvent(s, ly.ev_fw, 0)); w2 = ly.fw.img; wmax = ly.fw.words; }
      else if ((r = conv3x3_raw_h2_weights(ctx, P + ly.o_wf, ly.Cin_p, ly.Cout_p, &wsc, &w2, &wmax)) != AGZ_OK) return r;
      ConvDmaArgs ca{};
      ca.xh = ly.xh2; ca.xl = ly.xh2 + n_x_fwd; ca.w2 = (const _Float16*)w2; ca.y = ly.z; ca.x_amax = amax_words + 2 * l + 1; ca.w_amax = wmax;
      ca.g = g; ca.Cin = ly.Cin_p; ca.Ntot = ly.Cout_p;
      if (conv3_layer(l)) {
        ca.n_mtiles = ceil_div(g.M, 256); ca.n_ntiles = ly.Cout_p / 128;
        switch (conv3_mode) {
          case 1: hipLaunchKernelGGL(k_conv_h2dma3<1>, dim3(ca.n_mtiles * ca.n_ntiles), dim3(256), 0, s, ca); break;
          case 2: hipLaunchKernelGGL(k_conv_h2dma3<2>, dim3(ca.n_mtiles * ca.n_ntiles), dim3(256), 0, s, ca); break;
          case 3: hipLaunchKernelGGL(k_conv_h2dma3<3>, dim3(ca.n_mtiles * ca.n_ntiles), dim3(256), 0, s, ca); break;
          default: hipLaunchKernelGGL(k_conv_h2dma3<0>, dim3(ca.n_mtiles * ca.n_ntiles), dim3(256), 0, s, ca); break;
        }
      } else {
        ca.n_mtiles = ceil_div(g.M, 128); ca.n_ntiles = ly.Cout_p / 256;
        hipLaunchKernelGGL(k_conv_h2dma, dim3(ca.n_mtiles * ca.n_ntiles), dim3(256), 0, s, ca);
      }
      ly.x_planes = true;
      r = AGZ_OK;
    } else if (use_h2_fwd(ly.Cin_p, ly.Cout_p)) {
      r = conv3x3_raw_h2(ctx, cur, P + ly.o_wf, ly.z, B, g.H, g.W, ly.Cin_p, ly.Cout_p, &wsc, xb_ready[l] ? board_words + (size_t)l * B : nullptr);
    } else if (use_x3(ly, ly.Cin_p, ly.Cout_p)) {
      if ((r = split_w3(ctx, P + ly.o_wf, ly.w3f, ly.Cout_p, ly.Cin_p)) != AGZ_OK) return r;
      r = conv3x3_raw_x3(ctx, cur, ly.w3f, ly.z, B, g.H, g.W, ly.Cin_p, ly.Cout_p);
    } else {
      r = conv3x3_raw(ctx, cur, P + ly.o_wf, ly.z, B, g.H, g.W, ly.Cin_p, ly.Cout_p);
    }
    if (r != AGZ_OK) return r;
    int C = ly.Cout_p;
    if (C % 4 == 0 && C <= 1024 && g.M >= 4096) {   // (small problems keep the two-pass form: nothing to gain, and it is the oracle's order)
      hipLaunchKernelGGL(k_bn_stats, dim3(nblk(g.M, 128)), dim3(256), 0, s, g, ly.z, C, acc, 128);   // (256 / 128 / 64 / 32 rows: 46 / 37 / 46 / 71 us at G19)
      hipLaunchKernelGGL(k_bn_fin2, dim3(nblk(C)), dim3(256), 0, s, acc, C, (double)g.M, conf.bn_eps, ly.mean, ly.inv);
    } else {
      hipLaunchKernelGGL(k_bn_sum, dim3(nblk(g.M, RPB)), dim3(std::min(C, 512)), 0, s, g, ly.z, C, (const float*)nullptr, acc, RPB);
      hipLaunchKernelGGL(k_bn_fin, dim3(nblk(C)), dim3(256), 0, s, acc, C, (double)g.M, conf.bn_eps, ly.mean, ly.inv, 0);
      hipLaunchKernelGGL(k_bn_sum, dim3(nblk(g.M, RPB)), dim3(std::min(C, 512)), 0, s, g, ly.z, C, (const float*)ly.mean, acc, RPB);
      hipLaunchKernelGGL(k_bn_fin, dim3(nblk(C)), dim3(256), 0, s, acc, C, (double)g.M, conf.bn_eps, ly.mean, ly.inv, 1);
    }
    x_amax_ready[l + 1] = 0; xb_ready[l + 1] = 0;
    if (wino && Kp % 4 == 0 && Kp / 4 <= 256 && 256 % (Kp / 4) == 0 && ly.nbr <= 2) {
      const int tpr = 256 / (Kp / 4), rpb = round_up(std::max(tpr, ceil_div(g.M, 2048)), tpr);
      const bool per_board = rpb <= g.HW;   // (a block's rows then lie in at most two boards)
      // the next layer's DMA convolution reads `out` as fp16 hi / lo planes: written here, under last step's range (k_split_h2p_cond checks)
      _Float16* nph = nullptr;
      if (l < L && dma_layer(l + 1) && layers[l + 1].Cin_p == Kp) {
        TLayer& nx = layers[l + 1];
        const size_t n_nx = (size_t)B * g.Hp * g.Wp * nx.Cin_p;
        if (!nx.xh2) { int r2 = alloc(&nx.xh2, 2 * n_nx); if (r2 != AGZ_OK) return r2; }
        nph = nx.xh2;
        planes_fused[l + 1] = 1;
      } else if (l < L) planes_fused[l + 1] = 0;
      hipLaunchKernelGGL(k_bn_apply_v, dim3(nblk(g.M, rpb)), dim3(256), 0, s, g, ly.z, P + ly.o_gamma, P + ly.o_beta, ly.mean, ly.inv, ly.out,
                         Kp, ly.nbr, rpb, amax_words + 2 * (l + 1) + 1, per_board ? board_words + (size_t)(l + 1) * B : nullptr,
                         nph, nph ? nph + (size_t)B * g.Hp * g.Wp * Kp : nullptr, (const unsigned*)(amax_prev + 2 * (l + 1) + 1));
      x_amax_ready[l + 1] = 1;
      xb_ready[l + 1] = per_board;
    } else {
      if (l < L) planes_fused[l + 1] = 0;
      hipLaunchKernelGGL(k_bn_apply, dim3(nblk((size_t)g.M * Kp)), dim3(256), 0, s, g, ly.z, P + ly.o_gamma, P + ly.o_beta, ly.mean,
                         ly.inv, ly.out, Kp, ly.nbr);
    }
    cur = ly.out;
  }
  // ---- heads forward
  HeadT h{};
  h.B = B; h.HW = g.HW; h.A = A; h.FC = FC; h.yh = yh; h.Wp = P + o_Wp; h.bp = P + o_bp; h.W1 = P + o_W1; h.b1 = P + o_b1;
  h.W2 = P + o_W2; h.b2 = P + o_b2; h.Pi = pi; h.V = v; h.logits = logits; h.hpre = hpre; h.o = o;
  h.dWp = G + o_Wp; h.dbp = G + o_bp; h.dW1 = G + o_W1; h.db1 = G + o_b1; h.dW2 = G + o_W2; h.db2 = G + o_b2; h.dyh = dyh; h.cost = cost;
  // (second form of the head kernels: see k_head_conv2 ... above; fast_heads = 0, agz_debug.h, keeps the first form for A/B)
  const int fcK = 2 * g.HW > FC ? 2 * g.HW : FC, fcJ = A > FC ? A : FC;
  const bool fh = fast_heads && (size_t)8 * (2 * g.HW) * 4 <= 60000 && (size_t)B * 8 * 4 <= 60000 && (size_t)8 * fcJ * 4 <= 60000 && fcK > 0;
  if (fh) {
    AGZ_HIP_TRY(hipMemsetAsync(head_acc, 0, 16 * sizeof(double), s));
    hipLaunchKernelGGL(k_head_conv2, dim3(std::min(nblk(g.M, 4), ctx->num_cus * 16)), dim3(256), 0, s, g, cur, P + o_hc, zh, Kp);
    hipLaunchKernelGGL(k_head_stats_part, dim3(std::min(nblk(g.M), ctx->num_cus)), dim3(256), 0, s, g, zh, head_acc);
    hipLaunchKernelGGL(k_head_stats_fin, dim3(1), dim3(64), 0, s, g, head_acc, conf.bn_eps, hmean, hinv);
    hipLaunchKernelGGL(k_head_apply, dim3(nblk((size_t)g.M * 3)), dim3(256), 0, s, g, zh, P + o_hg, P + o_hb, hmean, hinv, yh);
    hipLaunchKernelGGL(k_fc_fwd2, dim3(nblk(std::max(A, FC)), nblk(B, 8), 2), dim3(256), (size_t)8 * 2 * g.HW * sizeof(float), s, h);
    hipLaunchKernelGGL(k_value_out2, dim3(B), dim3(64), 0, s, h);
    hipLaunchKernelGGL(k_cost_part, dim3(std::min(nblk((size_t)B * A), 64)), dim3(256), 0, s, h, head_acc);
    hipLaunchKernelGGL(k_cost_fin, dim3(1), dim3(64), 0, s, h, head_acc);
  } else {
  hipLaunchKernelGGL(k_head_conv, dim3(nblk((size_t)g.M * 3)), dim3(256), 0, s, g, cur, P + o_hc, zh, Kp);
  hipLaunchKernelGGL(k_head_stats, dim3(3), dim3(256), 0, s, g, zh, conf.bn_eps, hmean, hinv);
  hipLaunchKernelGGL(k_head_apply, dim3(nblk((size_t)g.M * 3)), dim3(256), 0, s, g, zh, P + o_hg, P + o_hb, hmean, hinv, yh);
  hipLaunchKernelGGL(k_fc_fwd, dim3(nblk((size_t)B * A + (size_t)B * FC)), dim3(256), 0, s, h);
  hipLaunchKernelGGL(k_value_out, dim3(nblk(B)), dim3(256), 0, s, h);
  hipLaunchKernelGGL(k_cost, dim3(1), dim3(256), 0, s, h);
  }
  // ---- heads backward
  size_t n_fc = (size_t)2 * g.HW * A + (size_t)B * A + (size_t)g.HW * FC + (size_t)B * FC + FC + B + (size_t)B * 3 * g.HW;
  if (fh) {
    hipLaunchKernelGGL(k_fc_bwd_w, dim3(nblk(std::max(A, FC)), nblk(2 * g.HW, 8), 2), dim3(256), (size_t)B * 8 * sizeof(float), s, h);
    hipLaunchKernelGGL(k_fc_bwd_small, dim3(nblk((size_t)B * A + (size_t)B * FC + FC + B)), dim3(256), 0, s, h);
    hipLaunchKernelGGL(k_fc_bwd_y, dim3(nblk(2 * g.HW), nblk(B, 8), 2), dim3(256), (size_t)8 * fcJ * sizeof(float), s, h);
    hipLaunchKernelGGL(k_head_bn_bwd_a, dim3(std::min(nblk(g.M), ctx->num_cus)), dim3(256), 0, s, g, zh, yh, dyh, P + o_hg, hmean, hinv, G + o_hg, G + o_hb, dzh, head_acc);
    hipLaunchKernelGGL(k_head_bn_bwd_b, dim3(nblk((size_t)g.M * 3)), dim3(256), 0, s, g, zh, hmean, hinv, dzh, head_acc);
    hipLaunchKernelGGL(k_head_conv_bwd_w2, dim3(nblk(g.M, RPB)), dim3(256), 0, s, g, cur, dzh, G + o_hc, Kp, RPB);
  } else {
  hipLaunchKernelGGL(k_fc_bwd, dim3(nblk(n_fc)), dim3(256), 0, s, h);
  hipLaunchKernelGGL(k_head_bn_bwd, dim3(3), dim3(256), 0, s, g, zh, yh, dyh, P + o_hg, hmean, hinv, G + o_hg, G + o_hb, dzh);
  hipLaunchKernelGGL(k_head_conv_bwd_w, dim3(nblk(g.M, RPB)), dim3(256), 0, s, g, cur, dzh, G + o_hc, Kp, RPB);
  }
  float* dcur = dA;
  float* dnext = dB;
  if (fh) hipLaunchKernelGGL(k_head_conv_bwd_x2, dim3(nblk((size_t)g.M * (Kp / 4))), dim3(256), 0, s, g, dzh, P + o_hc, dcur, Kp);
  else hipLaunchKernelGGL(k_head_conv_bwd_x, dim3(nblk((size_t)g.M * Kp)), dim3(256), 0, s, g, dzh, P + o_hc, dcur, Kp);
  if (wino) {
    // the weight gradient's fp16 scratch at its largest layer, BEFORE the first slice goes to the reduction queue: an allocation failure
    // between two slices would leave the peers waiting in the collectives this rank never enters
    size_t need_dz = 0, need_x = 0;
    for (int l = 0; l <= L; l++) {
      need_dz = std::max(need_dz, (size_t)B * g.Hp * g.Wp * layers[l].Cout_p);
      need_x = std::max(need_x, (size_t)B * g.Hp * g.Wp * layers[l].Cin_p);
    }
    if (dz_h2_cap < need_dz) { if (dz_h2) hipFree(dz_h2); dz_h2 = nullptr; dz_h2_cap = 0; AGZ_HIP_TRY(hipMalloc(&dz_h2, need_dz * 4)); dz_h2_cap = need_dz; }
    if (x_h2_cap < need_x) { if (x_h2) hipFree(x_h2); x_h2 = nullptr; x_h2_cap = 0; AGZ_HIP_TRY(hipMalloc(&x_h2, need_x * 4)); x_h2_cap = need_x; }
  }
  if (on_slice) { int r = on_slice(o_hc, n_flat - o_hc, s); if (r != AGZ_OK) return r; }   // the heads' gradients are final
  // ---- tower backward
  { int r = side_stream(); if (r != AGZ_OK) return r; }
  hipStream_t sw = one_stream ? s : wg_stream;
  if (wino && hoist_w && !one_stream) AGZ_HIP_TRY(hipStreamWaitEvent(s, ev_bw, 0));   // the data gradient's weight images (prep_weights; long done)
  for (int l = L; l >= 0; l--) {
    TLayer& ly = layers[l];
    int C = ly.Cout_p;
    const float* xin = l == 0 ? x0 : layers[l - 1].out;
    double* s1 = acc_b + (size_t)l * 2048; double* s2 = s1 + 1024;
    unsigned* wg_amax = amax_words + 2 * l;
    float* dz = l == 0 ? this->dz0 : this->dz;  // (different pixel strides: keep the [pix][2Kp] buffer's zero halo intact)
    if (sw != s && l < L) AGZ_HIP_TRY(hipStreamWaitEvent(s, ev_split, 0));   // the previous layer's weight gradient has taken its copy of dz
    hipLaunchKernelGGL(k_bn_bwd1, dim3(nblk(g.M, RPB)), dim3(std::min(C, 512)), 0, s, g, ly.z, P + ly.o_gamma, P + ly.o_beta, ly.mean, ly.inv,
                       ly.out, dcur, G + ly.o_gamma, G + ly.o_beta, dz, s1, s2, Kp, ly.nbr, RPB, fuse_lr);
    bool dz_amax_ready = false, dzb_ready = false;
    if (wino && C % 4 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0) {
      const int tpr = 256 / (C / 4), rpb = round_up(std::max(tpr, ceil_div(g.M, 2048)), tpr);
      dzb_ready = rpb <= g.HW;
      hipLaunchKernelGGL(k_bn_bwd2_v, dim3(nblk(g.M, rpb)), dim3(256), 0, s, g, ly.z, ly.mean, ly.inv, dz, s1, s2, C, rpb, wg_amax,
                         dzb_ready ? board_words + (size_t)(L + 2 + l) * B : nullptr);
      dz_amax_ready = true;
    } else
      hipLaunchKernelGGL(k_bn_bwd2, dim3(nblk((size_t)g.M * C)), dim3(256), 0, s, g, ly.z, ly.mean, ly.inv, dz, s1, s2, C);
    // weight gradient (on its own stream from here)
    if (sw != s) { AGZ_HIP_TRY(hipEventRecord(ev_dz, s)); AGZ_HIP_TRY(hipStreamWaitEvent(sw, ev_dz, 0)); }
    bool split_recorded = false;
    WgArgs wa{};
    wa.dz = dz; wa.x = xin; wa.dw = G + ly.o_wf; wa.g = g; wa.N = C; wa.Cin = ly.Cin_p; wa.rows_per_block = 2048;   // rows of the reduction per workgroup
    wa.n_tiles = ceil_div(C, 128); wa.c_tiles = ceil_div(ly.Cin_p, 128);
    int chunks = ceil_div(g.M, wa.rows_per_block);
    wa.n_chunks = chunks;
    const unsigned wg_grid = (unsigned)(wa.n_tiles * wa.c_tiles * 9 * round_up(chunks, 8));   // k_wgrad_x3 / _h2: XCD-aware order
    const bool chip_full = x3_force || (size_t)wa.n_tiles * wa.c_tiles * 9 * chunks >= (size_t)ctx->num_cus;
    const bool off31 = (size_t)B * g.Hp * g.Wp * (size_t)std::max(C, ly.Cin_p) * 4 < ((size_t)1 << 31);   // byte offsets of the buffer loads
    if (wino && chip_full && off31 && C % 4 == 0 && ly.Cin_p % 4 == 0) {
      // fp16x2 products: range + split pass over dz and x (once per layer, not once per tap workgroup), then the GEMM
      const size_t n_dz = (size_t)B * g.Hp * g.Wp * C, n_x = (size_t)B * g.Hp * g.Wp * ly.Cin_p;
      if (dz_h2_cap < n_dz) { if (dz_h2) hipFree(dz_h2); dz_h2 = nullptr; dz_h2_cap = 0; AGZ_HIP_TRY(hipMalloc(&dz_h2, n_dz * 4)); dz_h2_cap = n_dz; }
      if (x_h2_cap < n_x) { if (x_h2) hipFree(x_h2); x_h2 = nullptr; x_h2_cap = 0; AGZ_HIP_TRY(hipMalloc(&x_h2, n_x * 4)); x_h2_cap = n_x; }
      const unsigned gs = (unsigned)std::min<size_t>(nblk(n_dz / 4), (size_t)ctx->num_cus * 8);
      if (!dz_amax_ready) hipLaunchKernelGGL(k_absmax, dim3(gs), dim3(256), 0, sw, dz, n_dz / 4, wg_amax);
      if (!x_amax_ready[l]) hipLaunchKernelGGL(k_absmax, dim3(gs), dim3(256), 0, sw, xin, n_x / 4, wg_amax + 1);
      if (g.W >= 16 && C % 8 == 0 && ly.Cin_p % 8 == 0) {
        // three taps per workgroup from hi / lo fp16 planes (k_wgrad_h2t3)
        _Float16* dzh = (_Float16*)dz_h2; _Float16* dzl = dzh + n_dz;
        _Float16* xh = ly.x_planes ? ly.xh2 : (_Float16*)x_h2; _Float16* xl = xh + n_x;   // (the forward pass's planes of this input, if it made them)
        hipLaunchKernelGGL(k_split_h2p, dim3(gs), dim3(256), 0, sw, dz, dzh, dzl, n_dz / 4, wg_amax);
        if (sw != s) { AGZ_HIP_TRY(hipEventRecord(ev_split, sw)); split_recorded = true; }
        if (!ly.x_planes) hipLaunchKernelGGL(k_split_h2p, dim3(gs), dim3(256), 0, sw, xin, xh, xl, n_x / 4, wg_amax + 1);
        WgH2t3Args w3{};
        w3.dzh = dzh; w3.dzl = dzl; w3.xh = xh; w3.xl = xl; w3.amax = wg_amax; w3.dw = wa.dw; w3.g = g; w3.N = wa.N; w3.Cin = wa.Cin;
        w3.n_tiles = wa.n_tiles; w3.c_tiles = wa.c_tiles; w3.steps_per_board = ceil_div(g.HW, 32);
        const int per_chunk3 = wa.n_tiles * wa.c_tiles * 3;
        // chunks of boards: a chunk's workgroups share an XCD (its L2 holds the chunk's rows), an XCD runs 2 workgroups on each of its
        // CUs; about two rounds of them (5 chunks x 24 workgroups on 64 slots at K = 256; 4 to 8 chunks per XCD measured within 5 %)
        const int slots = ctx->num_cus / 8 * 2;
        w3.n_chunks = std::min(B, 8 * std::max(1, 2 * slots / per_chunk3));
        hipLaunchKernelGGL(k_wgrad_h2t3, dim3((unsigned)(per_chunk3 * round_up(w3.n_chunks, 8))), dim3(256), (size_t)wg_pad_lds, sw, w3);
      } else {
        hipLaunchKernelGGL(k_split_h2, dim3(gs), dim3(256), 0, sw, dz, dz_h2, n_dz / 4, wg_amax);
        if (sw != s) { AGZ_HIP_TRY(hipEventRecord(ev_split, sw)); split_recorded = true; }
        hipLaunchKernelGGL(k_split_h2, dim3(gs), dim3(256), 0, sw, xin, x_h2, n_x / 4, wg_amax + 1);
        WgH2Args wh{};
        wh.dz2 = dz_h2; wh.x2 = x_h2; wh.amax = wg_amax; wh.dw = wa.dw; wh.g = g; wh.N = wa.N; wh.Cin = wa.Cin;
        wh.rows_per_block = wa.rows_per_block; wh.n_tiles = wa.n_tiles; wh.c_tiles = wa.c_tiles; wh.n_chunks = chunks;
        hipLaunchKernelGGL(k_wgrad_h2, dim3(wg_grid), dim3(256), 0, sw, wh);
      }
    }
    // bf16x3 mode: the weight gradient runs on the bf16 pipe as well
    else if (x3 && off31 && chip_full)
      hipLaunchKernelGGL(k_wgrad_x3, dim3(wg_grid), dim3(256), 0, sw, wa);
    else
      hipLaunchKernelGGL(k_wgrad, dim3(wa.n_tiles * wa.c_tiles * 9 * chunks), dim3(256), 0, sw, wa);
    if (sw != s && !split_recorded) AGZ_HIP_TRY(hipEventRecord(ev_split, sw));   // (kernels that read dz itself: after the weight gradient)
    // layer l's slice [filter | gamma | beta] is final once its weight gradient has run (sw: it started after this layer's BatchNorm backward)
    if (on_slice) { int r = on_slice(ly.o_wf, (l < L ? layers[l + 1].o_wf : o_hc) - ly.o_wf, sw); if (r != AGZ_OK) return r; }
    if (l > 0) {  // data gradient: the forward GEMM with flipped/transposed weights over the [a|b] channels of dz
      if (!ly.bw_ready) hipLaunchKernelGGL(k_make_wt, dim3(nblk((size_t)9 * C * ly.Cin_p)), dim3(256), 0, s, P + ly.o_wf, ly.wt, C, ly.Cin_p);
      int r;
      if (use_wino(C, ly.Cin_p)) {
        r = conv3x3_raw_wino_h2(ctx, dz, ly.wt, dnext, B, g.H, g.W, C, ly.Cin_p, &wsc, dzb_ready ? board_words + (size_t)(L + 2 + l) * B : nullptr,
                                ly.bw_ready ? &ly.bw : nullptr);
      } else if (use_x3(ly, C, ly.Cin_p)) {
        if ((r = split_w3(ctx, ly.wt, ly.w3t, ly.Cin_p, C)) != AGZ_OK) return r;
        r = conv3x3_raw_x3(ctx, dz, ly.w3t, dnext, B, g.H, g.W, C, ly.Cin_p);
      } else {
        r = conv3x3_raw(ctx, dz, ly.wt, dnext, B, g.H, g.W, C, ly.Cin_p);
      }
      if (r != AGZ_OK) return r;
      std::swap(dcur, dnext);
    }
  }
  if (sw != s) { AGZ_HIP_TRY(hipEventRecord(ev_join, sw)); AGZ_HIP_TRY(hipStreamWaitEvent(s, ev_join, 0)); }   // the step's stream carries everything again
  AGZ_HIP_TRY(hipGetLastError());
  return AGZ_OK;
}

// (comm.hip) the context a trainer was created on
agz_ctx* agz_trainer_ctx(const agz_trainer* t) { return t ? t->ctx : nullptr; }
void agz_trainer_set_slice_hook(agz_trainer* t, std::function<int(size_t, size_t, hipStream_t)> f) { t->on_slice = std::move(f); }
// the slices forward_backward_dev hands to on_slice, in the order it issues them: the heads, then layer L .. 0 ([filter | gamma | beta])
void agz_trainer_slices(const agz_trainer* t, std::vector<std::pair<size_t, size_t>>& out) {
  out.clear();
  out.emplace_back(t->o_hc, t->n_flat - t->o_hc);
  const int L = (int)t->layers.size() - 1;
  for (int l = L; l >= 0; l--) out.emplace_back(t->layers[l].o_wf, (l < L ? t->layers[l + 1].o_wf : t->o_hc) - t->layers[l].o_wf);
}

extern "C" {

int agz_trainer_create(agz_ctx* ctx, const agz_net_conf* c, agz_trainer** out) {
  AGZ_REQUIRE(ctx && c && out, AGZ_E_INVALID, "agz_trainer_create: NULL argument");
  AGZ_REQUIRE(c->K >= 1 && c->ActionSpace >= 3 && c->SharedLayers >= 0 && c->FC > 1 && c->BatchSize >= 1 && c->Features > 0,
              AGZ_E_INVALID, "agz_trainer_create: NNConf is not valid");
  AGZ_REQUIRE(c->Features <= 32, AGZ_E_UNSUPPORTED, "Features > 32 unsupported");
  AGZ_HIP_TRY(hipSetDevice(ctx->device));
  agz_trainer* t = new agz_trainer();
  t->ctx = ctx; t->conf = *c;
  t->K = c->K; t->Kp = round_up(c->K, 32); t->L = c->SharedLayers; t->A = c->ActionSpace; t->FC = c->FC; t->B = c->BatchSize; t->F = c->Features;
  AGZ_REQUIRE(2 * t->Kp <= 1024, AGZ_E_UNSUPPORTED, "K > 512 unsupported by the trainer");
  TGeo& g = t->g;
  g.B = t->B; g.H = c->Height; g.W = c->Width; g.HW = g.H * g.W; g.Hp = g.H + 2; g.Wp = g.W + 2; g.M = g.B * g.HW;
  const int K = t->K, Kp = t->Kp, HW = g.HW, B = t->B, H = g.H, W = g.W, A = t->A, FCn = t->FC, F = t->F;
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += (n + 3) & ~(size_t)3; return o; };
  t->layers.resize(t->L + 1);
  for (int l = 0; l <= t->L; l++) {
    TLayer& ly = t->layers[l];
    ly.nbr = l == 0 ? 1 : 2; ly.Cin_p = l == 0 ? t->Fp : Kp; ly.Cout_p = ly.nbr * Kp;
    ly.o_wf = take((size_t)9 * ly.Cout_p * ly.Cin_p);
    ly.o_gamma = take((size_t)g.M * ly.Cout_p);
    ly.o_beta = take((size_t)g.M * ly.Cout_p);
  }
  t->o_hc = take((size_t)3 * Kp); t->o_hg = take((size_t)B * 3 * HW); t->o_hb = take((size_t)B * 3 * HW);
  t->o_Wp = take((size_t)2 * HW * A); t->o_bp = take((size_t)B * A); t->o_W1 = take((size_t)HW * FCn); t->o_b1 = take((size_t)B * FCn);
  t->o_W2 = take(FCn); t->o_b2 = take(B);
  t->n_flat = off;
  // reference Model() order (include/agz.h)
  auto pr = [&](const std::string& nm, int kind, std::vector<int> shp, int layer, int sub) { t->prefs.push_back(TParamRef{nm, kind, shp, layer, sub}); };
  pr("FilterInit", 0, {K, F, 3, 3}, 0, 0); pr("Init_gamma", 1, {B, K, H, W}, 0, 1); pr("Init_beta", 1, {B, K, H, W}, 0, 2);
  for (int i = 0; i < t->L; i++) {
    std::string s = std::to_string(i);
    pr("FilterLayer1 of Shared Layer " + s, 0, {K, K, 3, 3}, i + 1, 0); pr("L1_" + s + "_gamma", 1, {B, K, H, W}, i + 1, 1); pr("L1_" + s + "_beta", 1, {B, K, H, W}, i + 1, 2);
    pr("FilterLayer2 of Shared Layer " + s, 0, {K, K, 3, 3}, i + 1, 10); pr("L2_" + s + "_gamma", 1, {B, K, H, W}, i + 1, 11); pr("L2_" + s + "_beta", 1, {B, K, H, W}, i + 1, 12);
  }
  pr("FilterPolicyHead", 0, {2, K, 1, 1}, -1, 0); pr("PolicyHead_gamma", 1, {B, 2, H, W}, -1, 1); pr("PolicyHead_beta", 1, {B, 2, H, W}, -1, 2);
  pr("Policy_w", 2, {2 * HW, A}, -1, 3); pr("Policy_b", 3, {B, A}, -1, 4);
  pr("FilterValueHead", 0, {1, K, 1, 1}, -2, 0); pr("ValueHead_gamma", 1, {B, 1, H, W}, -2, 1); pr("ValueHead_beta", 1, {B, 1, H, W}, -2, 2);
  pr("Value_w", 2, {HW, FCn}, -2, 3); pr("Value_b", 3, {B, FCn}, -2, 4);
  pr("ValueOutput_w", 2, {FCn, 1}, -2, 5); pr("ValueOutput_b", 3, {B, 1}, -2, 6);
  int r = AGZ_OK;
#define TAL(p, n) if ((r = t->alloc(&t->p, (size_t)(n))) != AGZ_OK) { agz_trainer_destroy(t); return r; }
  TAL(P, t->n_flat) TAL(G, t->n_flat)
  size_t px = (size_t)B * g.Hp * g.Wp;
  TAL(x0, px * t->Fp) TAL(dA, px * Kp) TAL(dB, px * Kp) TAL(dz, px * 2 * Kp) TAL(dz0, px * Kp) TAL(acc, 2048)   /* [2][1024] channel sums of the forward BatchNorm (self-clearing) */
  TAL(acc_b, (size_t)(t->L + 1) * 2048) TAL(amax_words, (size_t)(t->L + 2) * 2) TAL(zero_tab, (size_t)(t->L + 1) * 2)
  TAL(board_words, (size_t)2 * (t->L + 2) * B)
  t->x_amax_ready.assign(t->L + 2, 0); t->xb_ready.assign(t->L + 2, 0); t->planes_fused.assign(t->L + 2, 0);
  {
    std::vector<size_t> tab((size_t)(t->L + 1) * 2);
    for (int l = 0; l <= t->L; l++) {
      tab[l] = t->layers[l].o_wf;
      tab[t->L + 1 + l] = (size_t)9 * t->layers[l].Cout_p * t->layers[l].Cin_p;
    }
    AGZ_HIP_TRY(hipMemcpyAsync(t->zero_tab, tab.data(), tab.size() * sizeof(size_t), hipMemcpyHostToDevice, ctx->stream));   // (after alloc()'s clear, same stream)
    AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  for (int l = 0; l <= t->L; l++) {
    TLayer& ly = t->layers[l];
    if ((r = t->alloc(&ly.z, px * ly.Cout_p)) != AGZ_OK || (r = t->alloc(&ly.out, px * Kp)) != AGZ_OK ||
        (r = t->alloc(&ly.mean, (size_t)ly.Cout_p)) != AGZ_OK || (r = t->alloc(&ly.inv, (size_t)ly.Cout_p)) != AGZ_OK ||
        (r = t->alloc(&ly.wt, (size_t)9 * ly.Cout_p * ly.Cin_p)) != AGZ_OK) { agz_trainer_destroy(t); return r; }
  }
  TAL(zh, (size_t)g.M * 3) TAL(yh, (size_t)g.M * 3) TAL(dyh, (size_t)g.M * 3) TAL(dzh, (size_t)g.M * 3) TAL(hmean, 4) TAL(hinv, 4)
  TAL(logits, (size_t)B * A) TAL(hpre, (size_t)B * FCn) TAL(o, B) TAL(cost, 2) TAL(head_acc, 16) TAL(amax_prev, (size_t)(t->L + 2) * 2)
  TAL(d_planes, (size_t)B * F * HW) TAL(d_pi, (size_t)B * A) TAL(d_v, B)
#undef TAL
  AGZ_HIP_TRY(hipStreamSynchronize(ctx->stream));
  *out = t;
  return AGZ_OK;
}

void agz_trainer_destroy(agz_trainer* t) {
  if (!t) return;
  hipSetDevice(t->ctx->device);
  hipStreamSynchronize(t->ctx->stream);
  if (t->wg_stream) { hipStreamSynchronize(t->wg_stream); hipStreamDestroy(t->wg_stream); }
  for (auto& ly : t->layers) { agz::raw_weights_free(&ly.fw); agz::raw_weights_free(&ly.bw); if (ly.ev_fw) hipEventDestroy(ly.ev_fw); }
  if (t->ev_w0) hipEventDestroy(t->ev_w0);
  if (t->ev_bw) hipEventDestroy(t->ev_bw);
  if (t->ev_dz) hipEventDestroy(t->ev_dz);
  if (t->ev_split) hipEventDestroy(t->ev_split);
  if (t->ev_join) hipEventDestroy(t->ev_join);
  for (void* p : t->allocs) hipFree(p);
  wino_raw_scratch_free(&t->wsc);
  if (t->dz_h2) hipFree(t->dz_h2);
  if (t->x_h2) hipFree(t->x_h2);
  delete t;
}

int agz_trainer_num_params(const agz_trainer* t) { return t ? (int)t->prefs.size() : 0; }

static size_t pref_size(const TParamRef& p) { size_t n = 1; for (int d : p.shape) n *= (size_t)d; return n; }

int agz_trainer_param_info(const agz_trainer* t, int i, char* name, size_t cap, size_t* n) {
  AGZ_REQUIRE(t && i >= 0 && i < (int)t->prefs.size(), AGZ_E_INVALID, "agz_trainer_param_info: bad index");
  if (name && cap) { strncpy(name, t->prefs[i].name.c_str(), cap - 1); name[cap - 1] = 0; }
  if (n) *n = pref_size(t->prefs[i]);
  return AGZ_OK;
}

// reference layout <-> device layout for parameter i of flat buffer `buf` (P or G). dir 0: host -> device, 1: device -> host
// nb_sel > 0: only the first nb_sel batch rows of a conv-layer gamma/beta (agz_trainer_export needs row 0 only)
static int xfer_param(const agz_trainer* t, float* buf, int i, float* host, int dir, int nb_sel = 0) {
  const TParamRef& p = t->prefs[i];
  const int K = t->K, Kp = t->Kp, HW = t->g.HW, B = (nb_sel > 0 && nb_sel < t->B && t->prefs[i].layer >= 0 && (t->prefs[i].sub % 10) != 0) ? nb_sel : t->B, A = t->A, FCn = t->FC, F = t->F, Fp = t->Fp;
  hipStream_t s = t->ctx->stream;
  auto dev_rw = [&](size_t off, std::vector<float>& tmp) -> int {
    if (dir == 0) AGZ_HIP_TRY(hipMemcpyAsync(buf + off, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice, s));
    else AGZ_HIP_TRY(hipMemcpyAsync(tmp.data(), buf + off, tmp.size() * 4, hipMemcpyDeviceToHost, s));
    AGZ_HIP_TRY(hipStreamSynchronize(s));
    return AGZ_OK;
  };
  if (p.layer >= 0) {
    const TLayer& ly = t->layers[p.layer];
    int br = p.sub >= 10 ? 1 : 0, sub = p.sub % 10;
    int Cin = p.layer == 0 ? F : K, Cin_p = ly.Cin_p, C = ly.Cout_p;
    if (sub == 0) {  // filter [K][Cin][3][3] <-> wf[tap][br*Kp + o][ci]
      std::vector<float> tmp((size_t)9 * C * Cin_p);
      if (hipMemcpy(tmp.data(), buf + ly.o_wf, tmp.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { agz::set_error("xfer: D2H failed"); return AGZ_E_HIP; }
      for (int o = 0; o < K; o++) for (int ci = 0; ci < Cin; ci++) for (int tp = 0; tp < 9; tp++) {
        size_t di = ((size_t)tp * C + br * Kp + o) * Cin_p + ci, hi = ((size_t)o * Cin + ci) * 9 + tp;
        if (dir == 0) tmp[di] = host[hi]; else host[hi] = tmp[di];
      }
      if (dir == 0) return dev_rw(ly.o_wf, tmp);
      return AGZ_OK;
    }
    // gamma/beta [B][K][H][W] <-> [r = b*HW+p][br*Kp + c]: only this tensor's Kp-wide column block moves (strided 2-D
    // copy), the host-side transpose runs on all cores — a 19x19 / K=256 / B=256 trainer has 82 such tensors of 94 MB
    size_t off = sub == 1 ? ly.o_gamma : ly.o_beta;
    const size_t M = (size_t)B * HW;   // (B is nb_sel when only the leading rows are wanted)
    std::vector<float> tmp(M * Kp, 0.f);
    float* dcol = buf + off + (size_t)br * Kp;
    if (dir == 1 && hipMemcpy2D(tmp.data(), (size_t)Kp * 4, dcol, (size_t)C * 4, (size_t)Kp * 4, M, hipMemcpyDeviceToHost) != hipSuccess) {
      agz::set_error("xfer: D2H failed"); return AGZ_E_HIP;
    }
    auto work = [&](int b_lo, int b_hi) {
      for (int b = b_lo; b < b_hi; b++)
        for (int q = 0; q < HW; q++) {
          float* trow = tmp.data() + ((size_t)b * HW + q) * Kp;
          for (int c = 0; c < K; c++) {
            size_t hi = ((size_t)b * K + c) * HW + q;
            if (dir == 0) trow[c] = host[hi]; else host[hi] = trow[c];
          }
        }
    };
    unsigned nthr = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    if ((size_t)B * K * HW < (1u << 16) || nthr == 1) work(0, B);
    else {
      std::vector<std::thread> th;
      int per = (B + (int)nthr - 1) / (int)nthr;
      for (int lo = 0; lo < B; lo += per) th.emplace_back(work, lo, std::min(B, lo + per));
      for (auto& x : th) x.join();
    }
    if (dir == 0 && hipMemcpy2D(dcol, (size_t)C * 4, tmp.data(), (size_t)Kp * 4, (size_t)Kp * 4, M, hipMemcpyHostToDevice) != hipSuccess) {
      agz::set_error("xfer: H2D failed"); return AGZ_E_HIP;
    }
    return AGZ_OK;
  }
  // heads
  const bool pol = p.layer == -1;
  auto plain = [&](size_t off, size_t n) -> int {
    if (dir == 0) AGZ_HIP_TRY(hipMemcpy(buf + off, host, n * 4, hipMemcpyHostToDevice));
    else AGZ_HIP_TRY(hipMemcpy(host, buf + off, n * 4, hipMemcpyDeviceToHost));
    return AGZ_OK;
  };
  if (p.sub == 0) {  // 1x1 filter [nc][K] <-> hc[j][Kp]
    int nc = pol ? 2 : 1, j0 = pol ? 0 : 2;
    std::vector<float> tmp((size_t)3 * Kp);
    AGZ_HIP_TRY(hipMemcpy(tmp.data(), buf + t->o_hc, tmp.size() * 4, hipMemcpyDeviceToHost));
    for (int j = 0; j < nc; j++) for (int c = 0; c < K; c++) { if (dir == 0) tmp[(size_t)(j0 + j) * Kp + c] = host[(size_t)j * K + c]; else host[(size_t)j * K + c] = tmp[(size_t)(j0 + j) * Kp + c]; }
    if (dir == 0) AGZ_HIP_TRY(hipMemcpy(buf + t->o_hc, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
    return AGZ_OK;
  }
  if (p.sub == 1 || p.sub == 2) {  // head gamma/beta [B][nc][HW] <-> [B][3][HW]
    int nc = pol ? 2 : 1, j0 = pol ? 0 : 2;
    size_t off = p.sub == 1 ? t->o_hg : t->o_hb;
    std::vector<float> tmp((size_t)B * 3 * HW);
    AGZ_HIP_TRY(hipMemcpy(tmp.data(), buf + off, tmp.size() * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; b++) for (int j = 0; j < nc; j++) for (int q = 0; q < HW; q++) {
      size_t di = ((size_t)b * 3 + j0 + j) * HW + q, hi = ((size_t)b * nc + j) * HW + q;
      if (dir == 0) tmp[di] = host[hi]; else host[hi] = tmp[di];
    }
    if (dir == 0) AGZ_HIP_TRY(hipMemcpy(buf + off, tmp.data(), tmp.size() * 4, hipMemcpyHostToDevice));
    return AGZ_OK;
  }
  if (pol) return p.sub == 3 ? plain(t->o_Wp, (size_t)2 * HW * A) : plain(t->o_bp, (size_t)B * A);
  switch (p.sub) {
    case 3: return plain(t->o_W1, (size_t)HW * FCn);
    case 4: return plain(t->o_b1, (size_t)B * FCn);
    case 5: return plain(t->o_W2, FCn);
    default: return plain(t->o_b2, B);
  }
  (void)Fp;
}

int agz_trainer_set_param(agz_trainer* t, int i, const float* host, size_t n) {
  AGZ_REQUIRE(t && host && i >= 0 && i < (int)t->prefs.size(), AGZ_E_INVALID, "agz_trainer_set_param: bad argument");
  AGZ_REQUIRE(n == pref_size(t->prefs[i]), AGZ_E_INVALID, "agz_trainer_set_param(%s): need %zu floats, got %zu", t->prefs[i].name.c_str(), pref_size(t->prefs[i]), n);
  AGZ_HIP_TRY(hipSetDevice(t->ctx->device));
  AGZ_HIP_TRY(hipStreamSynchronize(t->ctx->stream));
  return xfer_param(t, t->P, i, const_cast<float*>(host), 0);
}
int agz_trainer_get_param(const agz_trainer* t, int i, float* host, size_t n) {
  AGZ_REQUIRE(t && host && i >= 0 && i < (int)t->prefs.size() && n >= pref_size(t->prefs[i]), AGZ_E_INVALID, "agz_trainer_get_param: bad argument");
  AGZ_HIP_TRY(hipSetDevice(t->ctx->device));
  AGZ_HIP_TRY(hipStreamSynchronize(t->ctx->stream));
  return xfer_param(t, t->P, i, host, 1);
}
int agz_trainer_get_grad(const agz_trainer* t, int i, float* host, size_t n) {
  AGZ_REQUIRE(t && host && i >= 0 && i < (int)t->prefs.size() && n >= pref_size(t->prefs[i]), AGZ_E_INVALID, "agz_trainer_get_grad: bad argument");
  AGZ_HIP_TRY(hipSetDevice(t->ctx->device));
  AGZ_HIP_TRY(hipStreamSynchronize(t->ctx->stream));
  return xfer_param(t, t->G, i, host, 1);
}

int agz_trainer_init_random(agz_trainer* t, uint64_t seed) {  // same recipe as agz_net_init_random over the FULL shapes
  AGZ_REQUIRE(t, AGZ_E_INVALID, "trainer is NULL");
  // SplitMix64 is counter based (state_n = seed + n * golden), so the sequential stream of the oracle's initialiser can be
  // generated in parallel: the batch-shaped gamma/beta of a 19x19, K=256, B=256 trainer are 1.9 G normals (37 s on one core).
  const uint64_t GOLD = 0x9E3779B97F4A7C15ull;
  uint64_t draws = 0;   // draws consumed so far
  unsigned nthr = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
  for (int i = 0; i < (int)t->prefs.size(); i++) {
    const TParamRef& p = t->prefs[i];
    std::vector<float> v(pref_size(p));
    double field = 1; for (size_t k = 2; k < p.shape.size(); k++) field *= p.shape[k];
    const double stdev = std::sqrt(2.0 / ((double)(p.shape[0] + p.shape[1]) * field));
    const size_t n = v.size();
    const uint64_t d0 = draws;
    if (p.kind == 0) {           // GlorotU: one draw per element
      const double lim = stdev * std::sqrt(3.0);
      auto work = [&](size_t lo, size_t hi) {
        SplitMix64 r(seed + (d0 + lo) * GOLD);
        for (size_t k = lo; k < hi; k++) v[k] = (float)((r.float64() * 2.0 - 1.0) * lim);
      };
      if (n < (1u << 16)) work(0, n);
      else {
        std::vector<std::thread> th;
        size_t per = (n + nthr - 1) / nthr;
        for (unsigned q = 0; q < nthr; q++) { size_t lo = q * per, hi = std::min(n, lo + per); if (lo < hi) th.emplace_back(work, lo, hi); }
        for (auto& x : th) x.join();
      }
      draws += n;
    } else if (p.kind == 1 || p.kind == 2) {   // GlorotN by Box-Muller: two draws per PAIR of elements
      const size_t pairs = (n + 1) / 2;
      auto work = [&](size_t lo, size_t hi) {   // pair indices
        SplitMix64 r(seed + (d0 + 2 * lo) * GOLD);
        for (size_t q = lo; q < hi; q++) {
          double u1 = 1.0 - r.float64(), u2 = r.float64();
          double rad = std::sqrt(-2.0 * std::log(u1)), th = 6.283185307179586476925 * u2;
          v[2 * q] = (float)(rad * std::cos(th) * stdev);
          if (2 * q + 1 < n) v[2 * q + 1] = (float)(rad * std::sin(th) * stdev);
        }
      };
      if (pairs < (1u << 15)) work(0, pairs);
      else {
        std::vector<std::thread> th;
        size_t per = (pairs + nthr - 1) / nthr;
        for (unsigned q = 0; q < nthr; q++) { size_t lo = q * per, hi = std::min(pairs, lo + per); if (lo < hi) th.emplace_back(work, lo, hi); }
        for (auto& x : th) x.join();
      }
      draws += 2 * pairs;
    }
    int rc = agz_trainer_set_param(t, i, v.data(), v.size());
    if (rc != AGZ_OK) return rc;
  }
  return AGZ_OK;
}

int agz_trainer_forward_backward(agz_trainer* t, const float* planes, const float* pi, const float* v, float* cost) {
  AGZ_REQUIRE(t && planes && pi && v, AGZ_E_INVALID, "agz_trainer_forward_backward: NULL argument");
  AGZ_HIP_TRY(hipSetDevice(t->ctx->device));
  hipStream_t s = t->ctx->stream;
  AGZ_HIP_TRY(hipMemcpyAsync(t->d_planes, planes, (size_t)t->B * t->F * t->g.HW * 4, hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipMemcpyAsync(t->d_pi, pi, (size_t)t->B * t->A * 4, hipMemcpyHostToDevice, s));
  AGZ_HIP_TRY(hipMemcpyAsync(t->d_v, v, (size_t)t->B * 4, hipMemcpyHostToDevice, s));
  int r = t->forward_backward_dev(t->d_planes, t->d_pi, t->d_v);
  if (r != AGZ_OK) return r;
  float c[2] = {0, 0};
  AGZ_HIP_TRY(hipMemcpyAsync(c, t->cost, 8, hipMemcpyDeviceToHost, s));
  AGZ_HIP_TRY(hipStreamSynchronize(s));
  if (cost) *cost = c[0] + c[1];
  return AGZ_OK;
}

// forward_backward on DEVICE buffers (a batch sliced out of agz_examples_tensors_dev): no staging copy
int agz_trainer_forward_backward_dev(agz_trainer* t, const float* planes_dev, const float* pi_dev, const float* v_dev, float* cost) {
  AGZ_REQUIRE(t && planes_dev && pi_dev && v_dev, AGZ_E_INVALID, "agz_trainer_forward_backward_dev: NULL argument");
  AGZ_HIP_TRY(hipSetDevice(t->ctx->device));
  int r = t->forward_backward_dev(planes_dev, pi_dev, v_dev);
  if (r != AGZ_OK) return r;
  if (cost) {
    float c[2] = {0, 0};
    AGZ_HIP_TRY(hipMemcpyAsync(c, t->cost, 8, hipMemcpyDeviceToHost, t->ctx->stream));
    AGZ_HIP_TRY(hipStreamSynchronize(t->ctx->stream));
    *cost = c[0] + c[1];
  }
  return AGZ_OK;
}

int agz_trainer_apply(agz_trainer* t, float lr, float grad_scale) {  // solver.Step (meta.go:40): w -= lr * grad
  AGZ_REQUIRE(t, AGZ_E_INVALID, "trainer is NULL");
  AGZ_HIP_TRY(hipSetDevice(t->ctx->device));
  if (t->fused_done) {
    // the backward that just ran stepped the tower's gamma / beta itself (k_bn_bwd1): the filters of every layer and the head region remain
    AGZ_REQUIRE(grad_scale == 1.0f, AGZ_E_STATE, "agz_trainer_apply: a fused step takes no gradient scale");
    for (const auto& ly : t->layers) {
      const size_t n = (size_t)9 * ly.Cout_p * ly.Cin_p;
      hipLaunchKernelGGL(k_axpy, dim3(nblk(n)), dim3(256), 0, t->ctx->stream, t->P + ly.o_wf, t->G + ly.o_wf, -lr, n);
    }
    const size_t nh = t->n_flat - t->o_hc;
    hipLaunchKernelGGL(k_axpy, dim3(nblk(nh)), dim3(256), 0, t->ctx->stream, t->P + t->o_hc, t->G + t->o_hc, -lr, nh);
    t->fused_done = false;
  } else {
    hipLaunchKernelGGL(k_axpy, dim3(nblk(t->n_flat)), dim3(256), 0, t->ctx->stream, t->P, t->G, -lr * grad_scale, t->n_flat);
  }
  AGZ_HIP_TRY(hipGetLastError());
  return AGZ_OK;
}

int agz_trainer_batch(agz_trainer* t, const float* planes, const float* pi, const float* v, float lr, float* cost) {
  AGZ_REQUIRE(t, AGZ_E_INVALID, "trainer is NULL");
  t->fuse_lr = lr;                                   // lr = 0 (a dry step): the plain path, gradients materialised
  int r = agz_trainer_forward_backward(t, planes, pi, v, cost);
  t->fused_done = r == AGZ_OK && t->fuse_lr != 0.f;
  t->fuse_lr = 0.f;
  if (r != AGZ_OK) return r;
  r = agz_trainer_apply(t, lr, 1.0f);
  if (r != AGZ_OK) return r;
  AGZ_HIP_TRY(hipStreamSynchronize(t->ctx->stream));
  return AGZ_OK;
}

// AGZ_COMPUTE_F32_MFMA (default); AGZ_COMPUTE_BF16X3: forward, data-gradient and weight-gradient GEMMs on the bf16 pipe;
// AGZ_COMPUTE_WINO_H2: the data-gradient convolutions of the dual blocks through the Winograd fp16x2 path (weights transformed on
// the device every step), everything else as in BF16X3.  Same gradient tolerance against the oracle in all modes.
int agz_trainer_set_compute_mode(agz_trainer* t, int mode) {
  AGZ_REQUIRE(t, AGZ_E_INVALID, "trainer is NULL");
  const bool force = (mode & AGZ_COMPUTE_FORCE) != 0;
  mode &= ~AGZ_COMPUTE_FORCE;
  AGZ_REQUIRE(mode == AGZ_COMPUTE_F32_MFMA || mode == AGZ_COMPUTE_BF16X3 || mode == AGZ_COMPUTE_WINO_H2, AGZ_E_INVALID, "agz_trainer_set_compute_mode: mode %d not available for training", mode);
  AGZ_HIP_TRY(hipSetDevice(t->ctx->device));
  t->x3_force = force;
  if (mode == AGZ_COMPUTE_BF16X3 || mode == AGZ_COMPUTE_WINO_H2)
    for (auto& ly : t->layers)
      if (!ly.w3f) {
        int r = t->alloc(&ly.w3f, w3_elems(ly.Cout_p, ly.Cin_p));
        if (r == AGZ_OK) r = t->alloc(&ly.w3t, w3_elems(ly.Cin_p, ly.Cout_p));
        if (r != AGZ_OK) return r;
      }
  t->x3 = mode == AGZ_COMPUTE_BF16X3 || mode == AGZ_COMPUTE_WINO_H2;
  t->wino = mode == AGZ_COMPUTE_WINO_H2;
  return AGZ_OK;
}

int agz_trainer_set_dma_forward(agz_trainer* t, int on) {
  AGZ_REQUIRE(t, AGZ_E_INVALID, "trainer is NULL");
  t->dma_fwd = (on & 1) != 0;
  t->fast_heads = (on & 4) == 0;      // bit 2: the FIRST form of the head kernels (A/B)
  t->conv3_mode = (on >> 6) & 3;
  t->conv3 = (on & 32) == 0;          // bit 5: the forward DMA convolution with one tap per K step (k_conv_h2dma) (A/B)
  t->one_stream = (on & 16) != 0;     // bit 4: everything on the step's stream (diagnostic)
  t->hoist_w = (on & 8) == 0;         // bit 3: weight images per layer in line, not at the start of the step on the side stream (A/B)
  t->wg_pad_lds = (on & 256) ? 0 : 16384;   // bit 8: k_wgrad_h2t3 at two workgroups per CU as in round 5 (default: one, 73.7 KB static + 16 KB dynamic LDS)
  t->wg_low_prio = (on & 512) != 0;     // bit 9: the side stream at the lowest priority (before the first step)
  return AGZ_OK;
}

int agz_trainer_grads_dev(agz_trainer* t, float** dev_ptr, size_t* n_floats) {
  AGZ_REQUIRE(t && dev_ptr && n_floats, AGZ_E_INVALID, "NULL argument");
  *dev_ptr = t->G; *n_floats = t->n_flat;
  return AGZ_OK;
}

// the learn rate dual.Train builds its solver with: gorgonia.NewVanillaSolver(gorgonia.WithLearnRate(0.1)) (dualnet/meta.go:22)
static constexpr float DUAL_TRAIN_LR = 0.1f;

// dual.Train (dualnet/meta.go:16-54): iterations x batches of BatchSize rows; shuffleBatch (meta.go:57-102) after every
// iteration with the build's SplitMix64 (Fisher-Yates j = r.Intn(i+1) pattern).  Xs/policies/values are shuffled IN PLACE
// like the reference does.
int agz_train(agz_trainer* t, float* Xs, float* policies, float* values, int batches, int iterations, uint64_t seed, float* last_cost) {
  AGZ_REQUIRE(t && Xs && policies && values && batches >= 1 && iterations >= 0, AGZ_E_INVALID, "agz_train: bad argument");
  const size_t xs = (size_t)t->F * t->g.HW, ps = (size_t)t->A;
  const size_t n = (size_t)batches * t->B;
  SplitMix64 rng(seed);
  std::vector<float> tmp(std::max(xs, ps));
  float c = 0;
  for (int it = 0; it < iterations; it++) {
    for (int b = 0; b < batches; b++) {
      size_t s0 = (size_t)b * t->B;
      int r = agz_trainer_batch(t, Xs + s0 * xs, policies + s0 * ps, values + s0, DUAL_TRAIN_LR, &c);
      if (r != AGZ_OK) return r;
    }
    for (size_t i = 0; i < n; i++) {
      size_t j = (size_t)(rng.next() % (uint64_t)(i + 1));
      if (j == i) continue;
      memcpy(tmp.data(), Xs + i * xs, xs * 4); memcpy(Xs + i * xs, Xs + j * xs, xs * 4); memcpy(Xs + j * xs, tmp.data(), xs * 4);
      memcpy(tmp.data(), policies + i * ps, ps * 4); memcpy(policies + i * ps, policies + j * ps, ps * 4); memcpy(policies + j * ps, tmp.data(), ps * 4);
      std::swap(values[i], values[j]);
    }
  }
  if (last_cost) *last_cost = c;
  return AGZ_OK;
}

// dual.Train over DEVICE tensors (Xs [rows,F,H,W], policies [rows,A], values [rows], rows = batches*BatchSize — e.g.
// agz_examples_tensors_dev).  Same loop and the same shuffleBatch stream as agz_train, but the shuffle permutes a row
// index (4 B/row) instead of swapping rows, each batch is gathered device to device, and the cost is read back once at
// the end — no host round trip of the examples and no per-batch synchronisation.  The device tensors are left unchanged.
int agz_train_dev(agz_trainer* t, const float* Xs_dev, const float* policies_dev, const float* values_dev, int batches, int iterations,
                  uint64_t seed, float* last_cost) {
  AGZ_REQUIRE(t && Xs_dev && policies_dev && values_dev && batches >= 1 && iterations >= 0, AGZ_E_INVALID, "agz_train_dev: bad argument");
  AGZ_HIP_TRY(hipSetDevice(t->ctx->device));
  hipStream_t s = t->ctx->stream;
  const size_t xs = (size_t)t->F * t->g.HW, ps = (size_t)t->A;
  const size_t n = (size_t)batches * t->B;
  std::vector<int32_t> perm(n);
  for (size_t i = 0; i < n; i++) perm[i] = (int32_t)i;
  int32_t* d_perm = nullptr;
  AGZ_HIP_TRY(hipMalloc(&d_perm, n * 4));
  SplitMix64 rng(seed);
  int rc = AGZ_OK;
  for (int it = 0; it < iterations && rc == AGZ_OK; it++) {
    hipError_t e = hipMemcpyAsync(d_perm, perm.data(), n * 4, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);   // perm is reshuffled on the host below
    if (e != hipSuccess) { agz::set_error("agz_train_dev: %s", hipGetErrorString(e)); rc = AGZ_E_HIP; break; }
    for (int b = 0; b < batches && rc == AGZ_OK; b++) {
      const int32_t* ib = d_perm + (size_t)b * t->B;
      hipLaunchKernelGGL(k_gather_rows_t, dim3(nblk((size_t)t->B * xs)), dim3(256), 0, s, Xs_dev, ib, t->d_planes, (int)xs, (size_t)t->B * xs);
      hipLaunchKernelGGL(k_gather_rows_t, dim3(nblk((size_t)t->B * ps)), dim3(256), 0, s, policies_dev, ib, t->d_pi, (int)ps, (size_t)t->B * ps);
      hipLaunchKernelGGL(k_gather_rows_t, dim3(nblk((size_t)t->B)), dim3(256), 0, s, values_dev, ib, t->d_v, 1, (size_t)t->B);
      t->fuse_lr = DUAL_TRAIN_LR;
      rc = t->forward_backward_dev(t->d_planes, t->d_pi, t->d_v);
      t->fused_done = rc == AGZ_OK;
      t->fuse_lr = 0.f;
      if (rc == AGZ_OK) rc = agz_trainer_apply(t, DUAL_TRAIN_LR, 1.0f);
    }
    for (size_t i = 0; i < n; i++) {  // shuffleBatch (meta.go:57-102) on the row index
      size_t j = (size_t)(rng.next() % (uint64_t)(i + 1));
      std::swap(perm[i], perm[j]);
    }
  }
  float c[2] = {0, 0};
  hipError_t e = hipMemcpyAsync(c, t->cost, 8, hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  hipFree(d_perm);
  if (rc != AGZ_OK) return rc;
  AGZ_HIP_TRY(e);
  if (last_cost) *last_cost = iterations > 0 ? c[0] + c[1] : 0.f;
  return AGZ_OK;
}

// Checkpoint of the TRAINABLE network (AZ.Save / AZ.Load, agogo.go:175-209, for the side that keeps learning): every
// learnable in its full batch-shaped form.  File: "AGZTRN01", agz_net_conf, count, then per tensor {n, n floats}.
int agz_trainer_save(const agz_trainer* t, const char* path) {
  AGZ_REQUIRE(t && path, AGZ_E_INVALID, "agz_trainer_save: NULL argument");
  FILE* f = fopen(path, "wb");
  AGZ_REQUIRE(f, AGZ_E_INVALID, "agz_trainer_save: cannot open %s", path);
  bool ok = fwrite("AGZTRN01", 1, 8, f) == 8 && fwrite(&t->conf, sizeof(t->conf), 1, f) == 1;
  uint64_t np = t->prefs.size();
  ok = ok && fwrite(&np, 8, 1, f) == 1;
  for (int i = 0; ok && i < (int)t->prefs.size(); i++) {
    std::vector<float> v(pref_size(t->prefs[i]));
    if (agz_trainer_get_param(t, i, v.data(), v.size()) != AGZ_OK) { fclose(f); return AGZ_E_HIP; }
    uint64_t cnt = v.size();
    ok = fwrite(&cnt, 8, 1, f) == 1 && fwrite(v.data(), 4, cnt, f) == cnt;
  }
  ok = (fclose(f) == 0) && ok;
  AGZ_REQUIRE(ok, AGZ_E_INVALID, "agz_trainer_save: write to %s failed", path);
  return AGZ_OK;
}

int agz_trainer_load(agz_trainer* t, const char* path) {
  AGZ_REQUIRE(t && path, AGZ_E_INVALID, "agz_trainer_load: NULL argument");
  FILE* f = fopen(path, "rb");
  AGZ_REQUIRE(f, AGZ_E_INVALID, "agz_trainer_load: cannot open %s", path);
  char magic[8];
  agz_net_conf c;
  uint64_t np = 0;
  bool ok = fread(magic, 1, 8, f) == 8 && memcmp(magic, "AGZTRN01", 8) == 0 && fread(&c, sizeof(c), 1, f) == 1 && fread(&np, 8, 1, f) == 1;
  if (ok) ok = memcmp(&c, &t->conf, sizeof(c)) == 0 && np == t->prefs.size();
  if (!ok) { fclose(f); agz::set_error("agz_trainer_load: %s is not a checkpoint of this trainer configuration", path); return AGZ_E_INVALID; }
  for (int i = 0; ok && i < (int)t->prefs.size(); i++) {
    uint64_t cnt = 0;
    std::vector<float> v(pref_size(t->prefs[i]));
    ok = fread(&cnt, 8, 1, f) == 1 && cnt == v.size() && fread(v.data(), 4, cnt, f) == cnt;
    if (ok && agz_trainer_set_param(t, i, v.data(), v.size()) != AGZ_OK) { fclose(f); return AGZ_E_HIP; }
  }
  fclose(f);
  AGZ_REQUIRE(ok, AGZ_E_INVALID, "agz_trainer_load: %s is truncated or mismatched", path);
  return AGZ_OK;
}

// the copy loop of dual.Infer (meta.go:141-146): row 0 of every learnable -> the inference net, then commit
int agz_trainer_export(const agz_trainer* t, agz_net* net) {
  AGZ_REQUIRE(t && net, AGZ_E_INVALID, "NULL argument");
  AGZ_REQUIRE((int)t->prefs.size() == agz_net_num_params(net), AGZ_E_INVALID, "agz_trainer_export: network shapes differ");
  AGZ_HIP_TRY(hipSetDevice(t->ctx->device));
  AGZ_HIP_TRY(hipStreamSynchronize(t->ctx->stream));
  for (int i = 0; i < (int)t->prefs.size(); i++) {
    size_t want = 0;
    int r = agz_net_param_info(net, i, nullptr, 0, &want);
    if (r != AGZ_OK) return r;
    const TParamRef& p = t->prefs[i];
    const bool conv_bn = p.layer >= 0 && (p.sub % 10) != 0;   // batch-shaped [B][K][H][W]: only row 0 is transferred
    std::vector<float> v(conv_bn ? want : pref_size(p));
    r = xfer_param(t, t->P, i, v.data(), 1, conv_bn ? 1 : 0);
    if (r != AGZ_OK) return r;
    AGZ_REQUIRE(v.size() >= want, AGZ_E_INVALID, "agz_trainer_export: parameter %d too small", i);
    r = agz_net_set_param(net, i, v.data(), v.size());  // takes row 0 of batch-shaped tensors
    if (r != AGZ_OK) return r;
  }
  return agz_net_commit(net);
}

}  // extern "C"
