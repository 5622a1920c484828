// Dual-block 3x3 convolution as Winograd F(4x4, 3x3) with the 36 transform-domain GEMMs on the bf16 matrix pipe
// (bf16x3 products, conv_x3.hpp) — AGZ_COMPUTE_WINO, opt-in.
//
// The bf16x3 direct kernel sits at the power ceiling of the matrix pipe: the lever left is fewer matrix
// instructions.  F(4x4,3x3) computes a 4x4 output tile from a 6x6 input tile with 36 multiplies per (cin, cout) pair
// instead of 144: on a 19x19 board (5x5 tiles = 20x20 outputs) 900 GEMM rows per board and channel pair instead of
// 3249 — 3.6x fewer MFMAs.  Y = At [ (G g Gt) (.) (Bt d B) ] A  (Lavin & Gray 2016; transform matrices below).
//
// Three kernels per layer, intermediates in HBM / Infinity Cache:
//   wino_in_kernel    x (padded NHWC fp32)           -> V[36][T][C]      fp32   (T = boards x tiles per board)
//   wino_gemm_kernel  V[pos] (T x C) * U3[pos] (C x 2K, bf16x3 pieces, pre-transformed at commit) -> M[36][T][2K] fp32
//   wino_out_kernel   M -> At M A, BN(scale,shift)+ReLU on both branches, add, ReLU -> y (padded NHWC fp32)
// Algorithmic HBM bytes per board and layer at K=256, 19x19: x 0.45 MB + V 2 x 0.92 MB + M 2 x 1.84 MB + y 0.45 MB
// = 6.4 MB (the direct kernel moves 0.9 MB): the path is bandwidth-heavy by construction, 512 boards = 3.3 GB per layer.
//
// Numerics: transforms in fp32 (Bt and At have small integer entries; G g Gt is evaluated in double at commit and rounded
// once), products fp32-grade (bf16x3), fp32 accumulation.  Measured on post-ReLU data (C=256, 19x19): rms error 1.0e-6 of
// the output rms against 2.2e-7 for direct fp32 accumulation — inside the stated network tolerance, but 5x the direct
// kernels' error: this is why the mode is opt-in and not the default arithmetic.
#pragma once
// (included by net.hip INSIDE namespace agz, after conv_x3.hpp, whose X3_* macros are still defined here)

struct WinoArgs {
  const float* x;            // layer input, board 0 of this chunk
  float* V;                  // [36][T][C]
  const unsigned short* U3;  // [36][C/16][3][Ntot][16] bf16 pieces
  float* Mb;                 // [36][T][Ntot]
  const void* ep;            // float4 {sa,ta,sb,tb} [HW][Cout_p]
  float* y;                  // layer output, board 0 of this chunk
  int B, H, W, Hp, Wp, C, Cout_p, Ntot;
  int nty, ntx, TPB, T;
  int n_mtiles, n_ntiles;
};

// Bt of F(4x4,3x3): rows of the 6x6 input transform (applied to columns, then to rows)
__device__ __forceinline__ void wino_bt6(const float d[6], float o[6]) {
  o[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  o[1] = -4.f * d[1] - 4.f * d[2] + d[3] + d[4];
  o[2] = 4.f * d[1] - 4.f * d[2] - d[3] + d[4];
  o[3] = -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
  o[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
  o[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
// At of F(4x4,3x3): 6 transform-domain values -> 4 outputs
__device__ __forceinline__ void wino_at4(const float m[6], float o[4]) {
  o[0] = m[0] + m[1] + m[2] + m[3] + m[4];
  o[1] = m[1] - m[2] + 2.f * m[3] - 2.f * m[4];
  o[2] = m[1] + m[2] + 4.f * m[3] + 4.f * m[4];
  o[3] = m[1] - m[2] + 8.f * m[3] - 8.f * m[4] + m[5];
}

// One thread per (tile, channel pair): 36 float2 loads (a wave reads 512 contiguous bytes per pixel), Bt d B, 36 float2 stores.
__global__ __launch_bounds__(256) void wino_in_kernel(WinoArgs a) {
  const int C2 = a.C >> 1;
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (size_t)a.T * C2) return;
  const int c2 = (int)(g % C2);
  const int t = (int)(g / C2);
  const int b = t / a.TPB, tt = t - b * a.TPB;
  const int ty = tt / a.ntx, tx = tt - ty * a.ntx;
  const float* xb = a.x + (size_t)b * a.Hp * a.Wp * a.C + 2 * c2;
  float tmx[6][6], tmy[6][6];   // Bt d : [xi][j], the two channels of this thread
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const int px = 4 * tx + j;             // padded column (image column 4*tx - 1 + j)
    float dx[6], dy[6], ox[6], oy[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
      const int py = 4 * ty + i;
      float2 v = make_float2(0.f, 0.f);
      if (py < a.Hp && px < a.Wp) v = *reinterpret_cast<const float2*>(xb + ((size_t)py * a.Wp + px) * a.C);
      dx[i] = v.x; dy[i] = v.y;
    }
    wino_bt6(dx, ox);
    wino_bt6(dy, oy);
#pragma unroll
    for (int i = 0; i < 6; i++) { tmx[i][j] = ox[i]; tmy[i][j] = oy[i]; }
  }
#pragma unroll
  for (int i = 0; i < 6; i++) {
    float ox[6], oy[6];
    wino_bt6(tmx[i], ox);
    wino_bt6(tmy[i], oy);
#pragma unroll
    for (int j = 0; j < 6; j++)
      *reinterpret_cast<float2*>(a.V + ((size_t)(i * 6 + j) * a.T + t) * a.C + 2 * c2) = make_float2(ox[j], oy[j]);
  }
}

// The 36 GEMMs M[pos] = V[pos] (T x C) * U[pos] (C x Ntot), bf16x3 products: the tile, staging, LDS layout and pipeline
// of conv3x3_x3_kernel with a plain K loop (16-channel chunks, no taps) and a raw store.  Workgroup order: position
// outermost and XCD-contiguous, so every XCD's L2 holds the weights of the one or two positions it is working on and
// the four column tiles of a row tile run back to back on the same XCD.
__global__ __launch_bounds__(256, 3) void wino_gemm_kernel(WinoArgs a) {
  constexpr int PIECE = 128 * 32;
  constexpr int STAGE = 6 * PIECE;
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * STAGE];

  const int per_pos = a.n_mtiles * a.n_ntiles;
  const int nblk = 36 * per_pos;
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
  const int NK = a.C >> 4;
  const unsigned piece_bytes = (unsigned)a.Ntot * 32u;
  const unsigned a_gbyte = (unsigned)((((size_t)pos * a.T + mrow) * a.C + shalf * 8) * 4);
  const unsigned b_gbyte = (unsigned)pos * (unsigned)NK * 3u * piece_bytes + (unsigned)nrow * 32u + (unsigned)shalf * 16u;
  const unsigned s_off = x3_lds_off(srow, shalf);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  unsigned fa[2], fb[2];
#pragma unroll
  for (int i = 0; i < 2; i++) fa[i] = x3_lds_off((wm * 2 + i) * 32 + (lane & 31), lane >> 5);
#pragma unroll
  for (int j = 0; j < 2; j++) fb[j] = x3_lds_off((wn * 2 + j) * 32 + (lane & 31), lane >> 5);

  float4 xa0, xa1, ya0, ya1;
  u32x4_t xb0, xb1, xb2, yb0, yb1, yb2;
  const char* xbase = reinterpret_cast<const char*>(a.V);
  const char* wbase = reinterpret_cast<const char*>(a.U3);
  int f_n = 0;
  unsigned xo_ = a_gbyte, wo_ = b_gbyte;
#undef X3_ADVANCE
#define X3_ADVANCE()                       \
  if (f_n + 1 < NK) {                      \
    f_n++;                                 \
    xo_ += 64u;                            \
    wo_ += 3u * piece_bytes;               \
  }

  X3_GLOAD(ya0, ya1, yb0, yb1, yb2)
  X3_GLOAD(xa0, xa1, xb0, xb1, xb2)
  X3_STORE_A(ya0, ya1, 0)
  X3_STORE_B(yb0, yb1, yb2, 0)
  __syncthreads();
  for (int it = 0; it < NK; it += 2) {
    X3_ITER(0, xa0, xa1, xb0, xb1, xb2, ya0, ya1, yb0, yb1, yb2)
    if (it + 1 < NK) X3_ITER(1, ya0, ya1, yb0, yb1, yb2, xa0, xa1, xb0, xb1, xb2)
  }

#pragma unroll
  for (int i = 0; i < 2; i++) {
#pragma unroll
    for (int r = 0; r < 16; r++) {
      const int row = (wm * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int m = m0 + row;
      if (m < a.T) {
        float* dst = a.Mb + ((size_t)pos * a.T + m) * a.Ntot;
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const int c = n0 + (wn * 2 + j) * 32 + (lane & 31);
          if (c < a.Ntot) dst[c] = acc[i][j][r];
        }
      }
    }
  }
}
// the shared X3_* staging / pipeline macros of conv_x3.hpp end here
#undef X3_ITER
#undef X3_QUAD
#undef X3_MF
#undef X3_SB
#undef X3_STORE_B
#undef X3_STORE_A
#undef X3_GLOAD
#undef X3_ADVANCE

// One thread per (tile, output channel): 2 x 36 loads (coalesced over channels), At M A on both branches, the dual-block
// epilogue, up to 16 stores (coalesced over channels).  Columns of M: [0,Cout_p) branch a, [Cout_p, 2*Cout_p) branch b.
__global__ __launch_bounds__(256) void wino_out_kernel(WinoArgs a) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (size_t)a.T * a.Cout_p) return;
  const int c = (int)(g % a.Cout_p);
  const int t = (int)(g / a.Cout_p);
  const int b = t / a.TPB, tt = t - b * a.TPB;
  const int ty = tt / a.ntx, tx = tt - ty * a.ntx;
  float Y[2][4][4];
#pragma unroll
  for (int br = 0; br < 2; br++) {
    float tm[4][6];   // At M : [k][nu]
#pragma unroll
    for (int nu = 0; nu < 6; nu++) {
      float m[6], o[4];
#pragma unroll
      for (int xi = 0; xi < 6; xi++) m[xi] = a.Mb[((size_t)(xi * 6 + nu) * a.T + t) * a.Ntot + br * a.Cout_p + c];
      wino_at4(m, o);
#pragma unroll
      for (int k = 0; k < 4; k++) tm[k][nu] = o[k];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) wino_at4(tm[k], Y[br][k]);
  }
  const float4* ep = reinterpret_cast<const float4*>(a.ep);
  float* yb = a.y + (size_t)b * a.Hp * a.Wp * a.Cout_p + c;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const int h = 4 * ty + k;
#pragma unroll
    for (int l = 0; l < 4; l++) {
      const int w = 4 * tx + l;
      if (h < a.H && w < a.W) {
        const float4 e = ep[(size_t)(h * a.W + w) * a.Cout_p + c];
        float va = Y[0][k][l] * e.x + e.y;
        float vb = Y[1][k][l] * e.z + e.w;
        va = va > 0.f ? va : 0.f;
        vb = vb > 0.f ? vb : 0.f;
        const float s = va + vb;
        yb[((size_t)(h + 1) * a.Wp + (w + 1)) * a.Cout_p] = s > 0.f ? s : 0.f;
      }
    }
  }
}

// Host: U[pos][n][ci] = (G g Gt)[xi][nu] of filter g = w[n][ci][3][3] (double, rounded once to fp32), split exactly into
// three bf16 pieces: u3[pos][ci/16][piece][n][ci%16].  get(n, ci, tap) returns the filter value (0 for padding).
template <typename Get>
static void wino_build_u3(std::vector<unsigned short>& u3, int Ntot, int C, Get get) {
  static const double G[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                 {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6},  {0, 0, 1}};
  const int NC = C / 16;
  u3.assign((size_t)36 * NC * 3 * Ntot * 16, 0);
  for (int n = 0; n < Ntot; n++)
    for (int ci = 0; ci < C; ci++) {
      double g[3][3], tg[6][3];
      bool any = false;
      for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { g[i][j] = get(n, ci, i * 3 + j); any = any || g[i][j] != 0.0; }
      if (!any) continue;
      for (int xi = 0; xi < 6; xi++) for (int j = 0; j < 3; j++) tg[xi][j] = G[xi][0] * g[0][j] + G[xi][1] * g[1][j] + G[xi][2] * g[2][j];
      for (int xi = 0; xi < 6; xi++) for (int nu = 0; nu < 6; nu++) {
        float v = (float)(tg[xi][0] * G[nu][0] + tg[xi][1] * G[nu][1] + tg[xi][2] * G[nu][2]);
        uint32_t u, hu, mu, lu; memcpy(&u, &v, 4);
        hu = u & 0xffff0000u; float hf; memcpy(&hf, &hu, 4);
        float r1 = v - hf; uint32_t ru; memcpy(&ru, &r1, 4);
        mu = ru & 0xffff0000u; float mf; memcpy(&mf, &mu, 4);
        float r2 = r1 - mf; memcpy(&lu, &r2, 4);
        const int pos = xi * 6 + nu;
        size_t base = ((((size_t)pos * NC + ci / 16) * 3) * Ntot + n) * 16 + (ci % 16);
        u3[base] = (unsigned short)(hu >> 16);
        u3[base + (size_t)Ntot * 16] = (unsigned short)(mu >> 16);
        u3[base + (size_t)2 * Ntot * 16] = (unsigned short)(lu >> 16);
      }
    }
}

// launches the selected stages for one chunk of boards; V / Mb sized by the caller
enum { WINO_IN = 1, WINO_GEMM = 2, WINO_OUT = 4 };
static void wino_launch(agz_ctx* ctx, WinoArgs& a, int stages) {
  a.nty = ceil_div(a.H, 4); a.ntx = ceil_div(a.W, 4); a.TPB = a.nty * a.ntx; a.T = a.B * a.TPB;
  a.n_mtiles = ceil_div(a.T, 128); a.n_ntiles = ceil_div(a.Ntot, 128);
  if (stages & WINO_IN) {
    ProfScope ps(ctx, AGZ_PROF_WINO_IN);
    const size_t n_in = (size_t)a.T * (a.C / 2);
    hipLaunchKernelGGL(wino_in_kernel, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, ctx->stream, a);
  }
  if (stages & WINO_GEMM) {
    ProfScope ps(ctx, AGZ_PROF_WINO_GEMM);
    const int nblk = 36 * a.n_mtiles * a.n_ntiles;
    hipLaunchKernelGGL(wino_gemm_kernel, dim3(nblk), dim3(256), 0, ctx->stream, a);
  }
  if (stages & WINO_OUT) {
    ProfScope ps(ctx, AGZ_PROF_WINO_OUT);
    const size_t n_out = (size_t)a.T * a.Cout_p;
    hipLaunchKernelGGL(wino_out_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, ctx->stream, a);
  }
}
