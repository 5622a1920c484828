// Winograd F(4x4,3x3) dual block with the output transform and the block epilogue INSIDE the GEMM kernel — round 3.
//
// conv_wino_h2.hpp's block moves 2.87 GB through HBM per 512-board launch where the layer's own input and output are 0.38 GB:
// 1.64 GB of it is the fp32 product M = V U written by the GEMM kernel and read back by the output-transform kernel.  Here M
// never exists in memory.  At M A is linear and separable in the position index (xi, nu):
//     Z[l]    += At[l][nu] * M[xi][nu]            after the K loop of position (xi, nu)          4 accumulators per output element
//     Y[k][l] += At[k][xi] * Z[l]                 after the six positions of row xi             16 accumulators per output element
// so a workgroup that owns a (tile block x channel block) and walks all 36 positions keeps 16 + 4 + 1 fp32 registers per
// (tile, GEMM column) and writes y — un-scaled, BN scale/shift, ReLU on both branches, add — and the per-tile maxima the next
// block's input transform reduces.  The register file bounds the workgroup's output tile: 64 tiles x 64 GEMM columns (32 output
// channels, both branches) on 8 waves x 8 accumulators x 21 = 168 registers of 256 (two waves per SIMD, so one wave's transform
// arithmetic runs under the other's MFMAs).  F(5x5,3x3) would need 25 + 5 + 1 registers per element and does not fit: the fused
// form runs F(4x4,3x3) (900 instead of 784 transform-domain rows per 19x19 board, +15 % MFMA work, no M traffic).
//
// Data path per K step (32 channels): the A fragments (64 tiles x 32 k, hi and lo piece) and the B fragments (64 columns x 32 k,
// hi and lo) = 16 KB arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`: no staging registers, no ds_write), each of the 8 waves
// fetching one 1 KB A fragment and one 1 KB B fragment in MFMA operand order (lane l = row l%16, k octet l/16), so a fragment
// read is one conflict-free 1 KB ds_read_b128.  Ring of NS stages, ONE raw s_barrier per K step, the DMA of step s + NS - 1
// issued right behind the barrier of step s, counted vmcnt (never 0 inside the loop).  V2 and U2 keep conv_wino_h2.hpp's layouts
// (V2 from wino_in_h2_kernel<4>, U2 from wino_build_u2<4>): a lane's 16 bytes are contiguous in both.
//
// Per workgroup: 36 positions x NK steps x (6 v_mfma_f32_16x16x32_f16 + 6 ds_read_b128 + 2 DMA) per wave.
// Roofline: MFMA (3 fp16 MFMAs per product): 2 * 36 * T * C * Ntot * 3 flop per launch against 2516.6 TF; its operand traffic
// (4 B per element and operand, 64 x 64 tile) is 85 B/clk/CU at the MFMA peak against ~56 B/clk/CU of L2 bandwidth: the kernel
// is bound by the L2 -> LDS path at about two thirds of the matrix peak.
#pragma once
// (included by net.hip INSIDE namespace agz, after conv_wino_h2.hpp)

typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// epilogue of both fused kernels: un-scale, BN scale/shift, ReLU per branch, add, store, per-tile maxima
__device__ __forceinline__ void wino_fused_epilogue(const WinoH2Args& h, unsigned char* lds, f32x4_t (&Y)[2][4][4], int lane, int tid, int wm, int wn,
                                                    int m0, int c0, int n_cb, int cbk) {
  const WinoArgs& a = h.w;
  // ---- epilogue: un-scale, BN scale/shift, ReLU per branch, add, store, per-tile maxima ----
  // lane: GEMM column j = lane % 16 = branch j / 8 of channel c; accumulator register r = tile row (lane / 16) * 4 + r of block i
  const int j = lane & 15, br = j >> 3, c = c0 + wn * 8 + (j & 7), g = lane >> 4;
  float* smax = reinterpret_cast<float*>(lds);                     // [4 column blocks][64 rows]
  const float2* ep2 = reinterpret_cast<const float2*>(a.ep);       // float4 {sa,ta,sb,tb} per (pixel, channel) = two float2
#pragma unroll
  for (int i = 0; i < 2; i++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = wm * 32 + i * 16 + g * 4 + r;
      const int t = m0 + row;
      const bool valid = t < a.T;
      const int tc = valid ? t : a.T - 1;
      const int b = tc / a.TPB, tt = tc - b * a.TPB;
      const int ty = tt / a.ntx, tx = tt - ty * a.ntx;
      float s_, unscale;
      wino_h2_scales(h.amax_in[b], WinoT<4>::VSHIFT, &s_, &unscale);
      unscale *= h.w_unscale;
      float* yb = a.y + (size_t)b * a.Hp * a.Wp * a.Cout_p + c;
      float mx = 0.f;
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int hh = 4 * ty + k, hc = hh < a.H ? hh : a.H - 1;
#pragma unroll
        for (int l = 0; l < 4; l++) {
          const int ww = 4 * tx + l, wc = ww < a.W ? ww : a.W - 1;
          const float2 e = ep2[((size_t)(hc * a.W + wc) * a.Cout_p + c) * 2 + br];
          float v = (Y[i][k][l][r] * unscale) * e.x + e.y;
          v = v > 0.f ? v : 0.f;
          v += __shfl_xor(v, 8, 64);                               // relu(a) + relu(b) >= 0 already
          if (valid && br == 0 && hh < a.H && ww < a.W) {
            yb[((size_t)(hh + 1) * a.Wp + (ww + 1)) * a.Cout_p] = v;
            mx = fmaxf(mx, v);
          }
        }
      }
      if (h.wave_max) {
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
        if (j == 0) smax[wn * 64 + row] = mx;
      }
    }
  }
  if (h.wave_max) {
    __syncthreads();
    if (tid < 64 && m0 + tid < a.T) {
      const float m = fmaxf(fmaxf(smax[tid], smax[64 + tid]), fmaxf(smax[128 + tid], smax[192 + tid]));
      h.wave_max[(size_t)(m0 + tid) * n_cb + cbk] = m;             // one word per (tile, 32 channels)
    }
  }
}

template <int NK, int NS>
__global__ __launch_bounds__(512, 2) void wino_fused4_h2_kernel(WinoH2Args h) {
  static_assert(NK % NS == 0 && NS >= 2 && NS - 1 <= NK, "stage index must be static inside a position");
  constexpr int STAGE = 16384;                     // 8 A fragments + 8 B fragments of 1 KB
  constexpr int NPOS = 36;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NS * STAGE];
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

  // LDS-DMA sources: wave w fetches A fragment (row block w/2, piece w%2) and B fragment (column block w/2, piece w%2)
  const __amdgpu_buffer_rsrc_t rV = h2_rsrc(a.V);
  const __amdgpu_buffer_rsrc_t rU = h2_rsrc(h.U2);
  unsigned voffA, voffB;
  {
    int t = m0 + (w >> 1) * 16 + (lane & 15);
    if (t >= a.T) t = a.T - 1;
    voffA = (unsigned)((((size_t)(t >> h.rsh) * h.rA + (size_t)(t & h.rmask)) * a.C) * 4) + (unsigned)(w & 1) * 64u + (unsigned)(lane >> 4) * 16u;
    const int j = lane & 15;
    const int n = (j >> 3) * a.Cout_p + c0 + (w >> 1) * 8 + (j & 7);
    voffB = (unsigned)(w & 1) * (unsigned)a.Ntot * 64u + (unsigned)n * 64u + (unsigned)(lane >> 4) * 16u;
  }
  const unsigned posA = h.rB * (unsigned)a.C * 4u;     // bytes from one position of V to the next
  const unsigned stepB = 2u * (unsigned)a.Ntot * 64u;  // bytes per K step of U2 ([pos][C/32][piece][Ntot][32])
  const unsigned dmaA = (unsigned)w * 1024u, dmaB = 8192u + (unsigned)w * 1024u;
  // fragment reads of this wave
  const unsigned rdA = (unsigned)(wm * 4) * 1024u + (unsigned)lane * 16u;      // + (i * 2 + piece) * 1024
  const unsigned rdB = 8192u + (unsigned)(wn * 2) * 1024u + (unsigned)lane * 16u;

#define WF_ISSUE(ST_, POS_, KC_)                                                                                              \
  {                                                                                                                           \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, (lds_ptr_t)(lds + (ST_) * STAGE + dmaA), 16, voffA,                           \
                                             (unsigned)(POS_) * posA + (unsigned)(KC_) * 128u, 0, 0);                         \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rU, (lds_ptr_t)(lds + (ST_) * STAGE + dmaB), 16, voffB,                           \
                                             ((unsigned)(POS_) * (unsigned)NK + (unsigned)(KC_)) * stepB, 0, 0);              \
  }

  f32x4_t Y[2][4][4], Z[2][4], acc[2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int l = 0; l < 4; l++) Y[i][k][l] = f32x4_t{0.f, 0.f, 0.f, 0.f};

#pragma unroll
  for (int s = 0; s < NS - 1; s++) WF_ISSUE(s, 0, s)

#pragma unroll 1
  for (int xi = 0; xi < 6; xi++) {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int l = 0; l < 4; l++) Z[i][l] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int nu = 0; nu < 6; nu++) {
      const int pos = xi * 6 + nu;
      acc[0] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      acc[1] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kc = 0; kc < NK; kc++) {
        // stage kc % NS holds step (pos, kc): this wave's own two DMAs of it have landed after the counted wait, everybody's
        // after the barrier — which every wave reaches only after its MFMAs of the previous step consumed their fragments,
        // so the stage of step s - 1 is free for the DMA of step s + NS - 1
        // (lgkmcnt(0): this wave's fragment reads of the previous step have RETURNED before it lets anyone overwrite their stage)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (NS - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        {
          const int kn = kc + NS - 1;
          int pn = kn < NK ? pos : pos + 1;
          int kk = kn < NK ? kn : kn - NK;
          if (pn >= NPOS) { pn = NPOS - 1; kk = NK - 1; }          // past the end: a harmless reload keeps the counts uniform
          WF_ISSUE((kc + NS - 1) % NS, pn, kk)
        }
        const unsigned char* sb = lds + (kc % NS) * STAGE;
        f16x8_t A_[2][2], B_[2];
#pragma unroll
        for (int p = 0; p < 2; p++) {
#pragma unroll
          for (int i = 0; i < 2; i++) A_[i][p] = *reinterpret_cast<const f16x8_t*>(sb + rdA + (unsigned)(i * 2 + p) * 1024u);
          B_[p] = *reinterpret_cast<const f16x8_t*>(sb + rdB + (unsigned)p * 1024u);
        }
#pragma unroll
        for (int pp = 0; pp < 3; pp++) {            // small terms first: lo*hi, hi*lo, hi*hi
          const int pa = pp == 0 ? 1 : 0, pb = pp == 1 ? 1 : 0;
#pragma unroll
          for (int i = 0; i < 2; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A_[i][pa], B_[pb], acc[i], 0, 0, 0);
        }
      }
      // Z[l] += At[l][nu] * M   (At columns: nu 0: 1,0,0,0 | 1: 1,1,1,1 | 2: 1,-1,1,-1 | 3: 1,2,4,8 | 4: 1,-2,4,-8 | 5: 0,0,0,1)
#pragma unroll
      for (int l = 0; l < 4; l++) {
        const float cf = WinoT<4>::AT[l][nu];
        if (cf != 0.f) {
#pragma unroll
          for (int i = 0; i < 2; i++) Z[i][l] += cf * acc[i];
        }
      }
    }
    // Y[k][l] += At[k][xi] * Z[l]
    const float ck[4] = {xi < 5 ? 1.f : 0.f,
                         xi == 1 ? 1.f : xi == 2 ? -1.f : xi == 3 ? 2.f : xi == 4 ? -2.f : 0.f,
                         (xi == 1 || xi == 2) ? 1.f : (xi == 3 || xi == 4) ? 4.f : 0.f,
                         xi == 1 ? 1.f : xi == 2 ? -1.f : xi == 3 ? 8.f : xi == 4 ? -8.f : xi == 5 ? 1.f : 0.f};
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int l = 0; l < 4; l++) Y[i][k][l] += ck[k] * Z[i][l];
  }
#undef WF_ISSUE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  wino_fused_epilogue(h, lds, Y, lane, tid, wm, wn, m0, c0, n_cb, cbk);
}

// Software-pipelined form of the same kernel.  Iteration s: [counted wait + barrier: stage s + 1 complete, everybody's reads of
// stage s returned] -> DMA of step s + NS into stage s -> fragment reads of step s + 1 into the OTHER register set -> the six
// MFMAs of step s from the set read one iteration ago (their reads had a whole MFMA group to return: no LDS wait in front of
// an MFMA) -> the transform arithmetic of the position that ended two iterations ago (two accumulator sets alternate by
// position parity, so these VALU instructions depend on nothing in flight and issue under the MFMAs).
template <int NK, int NS>
__global__ __launch_bounds__(512, 2) void wino_fused4p_h2_kernel(WinoH2Args h) {
  static_assert(NK % NS == 0 && NS >= 3 && NS <= NK && NK % 2 == 0, "static stage / register-set indices");
  constexpr int STAGE = 16384;
  constexpr int NPOS = 36;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NS * STAGE];
  const WinoArgs& a = h.w;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = w >> 2, wn = w & 3;
  const int n_cb = a.Cout_p >> 5;
  const int n_rb = (a.T + 63) >> 6;
  const int nblk = n_rb * n_cb;
  const int id = blockIdx.x;
  const int q_ = nblk >> 3, rr_ = nblk & 7, xcd = id & 7, slot = id >> 3;
  const int tile = (xcd < rr_ ? xcd * (q_ + 1) : rr_ * (q_ + 1) + (xcd - rr_) * q_) + slot;
  const int rbk = tile / n_cb, cbk = tile - rbk * n_cb;
  const int m0 = rbk * 64, c0 = cbk * 32;

  const __amdgpu_buffer_rsrc_t rV = h2_rsrc(a.V);
  const __amdgpu_buffer_rsrc_t rU = h2_rsrc(h.U2);
  unsigned voffA, voffB;
  {
    int t = m0 + (w >> 1) * 16 + (lane & 15);
    if (t >= a.T) t = a.T - 1;
    voffA = (unsigned)((((size_t)(t >> h.rsh) * h.rA + (size_t)(t & h.rmask)) * a.C) * 4) + (unsigned)(w & 1) * 64u + (unsigned)(lane >> 4) * 16u;
    const int j = lane & 15;
    const int n = (j >> 3) * a.Cout_p + c0 + (w >> 1) * 8 + (j & 7);
    voffB = (unsigned)(w & 1) * (unsigned)a.Ntot * 64u + (unsigned)n * 64u + (unsigned)(lane >> 4) * 16u;
  }
  const unsigned posA = h.rB * (unsigned)a.C * 4u;
  const unsigned stepB = 2u * (unsigned)a.Ntot * 64u;
  const unsigned dmaA = (unsigned)w * 1024u, dmaB = 8192u + (unsigned)w * 1024u;
  const unsigned rdA = (unsigned)(wm * 4) * 1024u + (unsigned)lane * 16u;
  const unsigned rdB = 8192u + (unsigned)(wn * 2) * 1024u + (unsigned)lane * 16u;

#define WF_ISSUE(ST_, POS_, KC_)                                                                                              \
  {                                                                                                                           \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rV, (lds_ptr_t)(lds + (ST_) * STAGE + dmaA), 16, voffA,                           \
                                             (unsigned)(POS_) * posA + (unsigned)(KC_) * 128u, 0, 0);                         \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rU, (lds_ptr_t)(lds + (ST_) * STAGE + dmaB), 16, voffB,                           \
                                             ((unsigned)(POS_) * (unsigned)NK + (unsigned)(KC_)) * stepB, 0, 0);              \
  }
#define WF_READ(SET_, ST_)                                                                                                    \
  {                                                                                                                           \
    const unsigned char* sb_ = lds + (ST_) * STAGE;                                                                           \
    _Pragma("unroll") for (int p = 0; p < 2; p++) {                                                                           \
      _Pragma("unroll") for (int i = 0; i < 2; i++)                                                                           \
        FA[SET_][i][p] = *reinterpret_cast<const f16x8_t*>(sb_ + rdA + (unsigned)(i * 2 + p) * 1024u);                        \
      FB[SET_][p] = *reinterpret_cast<const f16x8_t*>(sb_ + rdB + (unsigned)p * 1024u);                                       \
    }                                                                                                                         \
  }

  f32x4_t Y[2][4][4], Z[2][4], acc[2][2];
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
    acc[0][i] = zero4; acc[1][i] = zero4;
  }

  // prologue: DMA of steps 0 .. NS-2; stage 0 complete; DMA of step NS-1; fragments of step 0
#pragma unroll
  for (int s = 0; s < NS - 1; s++) WF_ISSUE(s, 0, s)
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NS - 2)) : "memory");
  __builtin_amdgcn_s_barrier();
  WF_ISSUE(NS - 1, 0, NS - 1)
  WF_READ(0, 0)

#pragma unroll 1
  for (int xi = 0; xi < 6; xi++) {
#pragma unroll
    for (int nu = 0; nu < 6; nu++) {
      const int pos = xi * 6 + nu;
#pragma unroll
      for (int kc = 0; kc < NK; kc++) {
        // iteration s = pos * NK + kc
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * (NS - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        {
          const int kn = kc + NS;                                   // DMA of step s + NS into stage s % NS (its reads returned: wait above)
          int pn = kn < NK ? pos : pos + 1;
          int kk = kn < NK ? kn : kn - NK;
          if (pn >= NPOS) { pn = NPOS - 1; kk = NK - 1; }           // past the end: a harmless reload keeps the counts uniform
          WF_ISSUE(kc % NS, pn, kk)
        }
        WF_READ((kc + 1) & 1, (kc + 1) % NS)                        // fragments of step s + 1 (past the end: stale bytes, never used)
        __builtin_amdgcn_sched_barrier(0);
        {
          constexpr int dummy = 0; (void)dummy;
          const int cs = kc & 1, par = nu & 1;
#pragma unroll
          for (int pp = 0; pp < 3; pp++) {          // small terms first: lo*hi, hi*lo, hi*hi
            const int pa = pp == 0 ? 1 : 0, pb = pp == 1 ? 1 : 0;
#pragma unroll
            for (int i = 0; i < 2; i++)
              acc[par][i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(FA[cs][i][pa], FB[cs][pb], (kc == 0 && pp == 0) ? zero4 : acc[par][i], 0, 0, 0);
          }
        }
        if (kc == 1) {
          // transform of the position that ended two iterations ago: (xi, nu - 1), or (xi - 1, 5) and the row update of xi - 1
          if (nu > 0) {
#pragma unroll
            for (int l = 0; l < 4; l++) {
              const float cf = WinoT<4>::AT[l][nu > 0 ? nu - 1 : 0];
              if (cf != 0.f) {
#pragma unroll
                for (int i = 0; i < 2; i++) Z[i][l] += cf * acc[(nu + 1) & 1][i];
              }
            }
          } else if (xi > 0) {
            const int xp = xi - 1;
#pragma unroll
            for (int i = 0; i < 2; i++) Z[i][3] += acc[1][i];       // At[.][5] = (0, 0, 0, 1)
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
      }
    }
  }
#undef WF_READ
#undef WF_ISSUE
  // last position (5, 5) and row 5: At[.][5] = (0, 0, 0, 1) in both directions
#pragma unroll
  for (int i = 0; i < 2; i++) {
    Z[i][3] += acc[1][i];
#pragma unroll
    for (int l = 0; l < 4; l++) Y[i][3][l] += Z[i][l];
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  wino_fused_epilogue(h, lds, Y, lane, tid, wm, wn, m0, c0, n_cb, cbk);
}

// K extents the fused kernel is instantiated for (32-channel steps)
static inline bool wino_fused_nk_ok(int C) {
  const int nk = C >> 5;
  return (C & 31) == 0 && (nk == 2 || nk == 4 || nk == 6 || nk == 8 || nk == 12 || nk == 16);
}

// input transform F(4x4,3x3) + the fused kernel for one dual block (h.w.V sized by wino_h2_rows(36, tiles); h.U2 the TM = 4 image)
static void wino_fused_launch(agz_ctx* ctx, WinoH2Args& h, hipStream_t st = nullptr) {
  if (!st) st = ctx->stream;
  WinoArgs& a = h.w;
  h.tm = 4; h.npos = 36;
  a.nty = ceil_div(a.H, 4); a.ntx = ceil_div(a.W, 4); a.TPB = a.nty * a.ntx; a.T = a.B * a.TPB;
  h.rsh = 7; h.rmask = 127; h.rA = 36u * 128u; h.rB = 128u;           // blocked layout [T/128][pos][128 rows]
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
    const int nk = a.C >> 5;
    // tuning knob AGZ_WINO_H2_FUSED_VAR: 0 = plain loop, 1 = software-pipelined (default), 2 = pipelined with an 8-stage ring
    static const int var = [] { const char* e = getenv("AGZ_WINO_H2_FUSED_VAR"); return e ? atoi(e) : 1; }();
#define AGZ_WF(NK_, NS_) hipLaunchKernelGGL((wino_fused4_h2_kernel<NK_, NS_>), grid, dim3(512), 0, st, h)
#define AGZ_WFP(NK_, NS_) hipLaunchKernelGGL((wino_fused4p_h2_kernel<NK_, NS_>), grid, dim3(512), 0, st, h)
    if (nk == 8 && var == 1) AGZ_WFP(8, 4);
    else if (nk == 8 && var == 2) AGZ_WFP(8, 8);
    else if (nk == 4 && var >= 1) AGZ_WFP(4, 4);
    else if (nk == 16 && var >= 1) AGZ_WFP(16, 4);
    else if (nk == 12 && var >= 1) AGZ_WFP(12, 4);
    else if (nk == 6 && var >= 1) AGZ_WFP(6, 3);
    else if (nk == 8) AGZ_WF(8, 4);
    else if (nk == 4) AGZ_WF(4, 4);
    else if (nk == 16) AGZ_WF(16, 4);
    else if (nk == 12) AGZ_WF(12, 4);
    else if (nk == 6) AGZ_WF(6, 3);
    else AGZ_WF(2, 2);
#undef AGZ_WFP
#undef AGZ_WF
  }
  if (h.wave_max && h.amax_out && !fuse)
    hipLaunchKernelGGL(wino_board_max_kernel, dim3(a.B), dim3(64), 0, st, h.wave_max, h.amax_out, h.wm_per_board);
}
