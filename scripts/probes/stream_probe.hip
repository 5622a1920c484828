// HBM ceiling probe for the Winograd output-transform access pattern (MI355X): how fast can 0.94 GB of M[36][T][512] fp32 be
// read (and 0.19 GB written) by kernels of different shapes?   hipcc --offload-arch=gfx950 -O3 stream_probe.hip -o stream_probe
//   P0  float4 grid-stride stream over the whole buffer (the copy-like ceiling)
//   P1  the out-kernel's mapping: one thread per (tile, channel), 72 dword loads at stride T*N, 16 dword stores
//   P2  P1 on a [T][36][N] layout (a tile's 72 KB contiguous)
//   P3  thread per (tile, channel pair): 72 dwordx2 loads
//   P4  P1 with two tiles per thread, the second tile's loads issued before the first is reduced
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int T = 12800, N = 512, C = 256;

__global__ __launch_bounds__(256) void p0(const float4* __restrict__ m, float* __restrict__ y, size_t n4) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { float4 v = m[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.f) y[0] = s;
}
template <bool TPOS>
__global__ __launch_bounds__(256) void p1(const float* __restrict__ m, float* __restrict__ y) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(g % C), t = (int)(g / C);
  float v[72];
#pragma unroll
  for (int br = 0; br < 2; br++)
#pragma unroll
    for (int p = 0; p < 36; p++) v[br * 36 + p] = TPOS ? m[((size_t)t * 36 + p) * N + br * C + c] : m[((size_t)p * T + t) * N + br * C + c];
  float s[16];
#pragma unroll
  for (int k = 0; k < 16; k++) s[k] = 0.f;
#pragma unroll
  for (int q = 0; q < 72; q++) s[q & 15] += v[q] * (float)(q + 1);
#pragma unroll
  for (int k = 0; k < 16; k++) y[((size_t)t * 16 + k) * C + c] = s[k];
}
__global__ __launch_bounds__(256) void p3(const float* __restrict__ m, float* __restrict__ y) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c2 = (int)(g % (C / 2)), t = (int)(g / (C / 2));
  float2 v[72];
#pragma unroll
  for (int br = 0; br < 2; br++)
#pragma unroll
    for (int p = 0; p < 36; p++) v[br * 36 + p] = *reinterpret_cast<const float2*>(m + ((size_t)p * T + t) * N + br * C + 2 * c2);
  float2 s[16];
#pragma unroll
  for (int k = 0; k < 16; k++) s[k] = make_float2(0.f, 0.f);
#pragma unroll
  for (int q = 0; q < 72; q++) { s[q & 15].x += v[q].x * (float)(q + 1); s[q & 15].y += v[q].y * (float)(q + 1); }
#pragma unroll
  for (int k = 0; k < 16; k++) *reinterpret_cast<float2*>(y + ((size_t)t * 16 + k) * C + 2 * c2) = s[k];
}
__global__ __launch_bounds__(256) void p4(const float* __restrict__ m, float* __restrict__ y) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(g % C), t0 = (int)(g / C) * 2;
  float v[2][72];
#pragma unroll
  for (int u = 0; u < 2; u++)
#pragma unroll
    for (int br = 0; br < 2; br++)
#pragma unroll
      for (int p = 0; p < 36; p++) v[u][br * 36 + p] = m[((size_t)p * T + t0 + u) * N + br * C + c];
#pragma unroll
  for (int u = 0; u < 2; u++) {
    float s[16];
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = 0.f;
#pragma unroll
    for (int q = 0; q < 72; q++) s[q & 15] += v[u][q] * (float)(q + 1);
#pragma unroll
    for (int k = 0; k < 16; k++) y[((size_t)(t0 + u) * 16 + k) * C + c] = s[k];
  }
}
// P5-P7: the output-transform kernel rebuilt piece by piece on P1's loads: + the 16 parameter vectors (L2-resident table),
// + the real At M A arithmetic and epilogue, + the per-board maximum (wave reduce + atomicMax)
__device__ __forceinline__ void at4(const float m[6], float o[4]) {
  o[0] = m[0] + m[1] + m[2] + m[3] + m[4];
  o[1] = m[1] - m[2] + 2.f * m[3] - 2.f * m[4];
  o[2] = m[1] + m[2] + 4.f * m[3] + 4.f * m[4];
  o[3] = m[1] - m[2] + 8.f * m[3] - 8.f * m[4] + m[5];
}
template <int LEVEL>
__global__ __launch_bounds__(256) void p5(const float* __restrict__ m, const float4* __restrict__ ep, float* __restrict__ y, unsigned* __restrict__ amax) {
  const size_t g = (size_t)blockIdx.x * 256 + threadIdx.x;
  const int c = (int)(g % C), t = (int)(g / C);
  const int b = t / 25, tt = t - b * 25, ty = tt / 5, tx = tt - ty * 5;
  float v[2][36];
#pragma unroll
  for (int br = 0; br < 2; br++)
#pragma unroll
    for (int p = 0; p < 36; p++) v[br][p] = m[((size_t)p * T + t) * N + br * C + c];
  float4 E[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    int hh = 4 * ty + (k >> 2), ww = 4 * tx + (k & 3);
    hh = hh < 19 ? hh : 18; ww = ww < 19 ? ww : 18;
    E[k] = ep[(size_t)(hh * 19 + ww) * C + c];
  }
  float Y[2][16];
  if (LEVEL >= 6) {
#pragma unroll
    for (int br = 0; br < 2; br++) {
      float tm[4][6];
#pragma unroll
      for (int nu = 0; nu < 6; nu++) {
        float mm[6], o[4];
#pragma unroll
        for (int xi = 0; xi < 6; xi++) mm[xi] = v[br][xi * 6 + nu];
        at4(mm, o);
#pragma unroll
        for (int k = 0; k < 4; k++) tm[k][nu] = o[k];
      }
#pragma unroll
      for (int k = 0; k < 4; k++) at4(tm[k], &Y[br][4 * k]);
    }
  } else {
#pragma unroll
    for (int br = 0; br < 2; br++)
#pragma unroll
      for (int k = 0; k < 16; k++) Y[br][k] = v[br][k] + v[br][k + 16] + v[br][(k + 32) % 36];
  }
  float mx = 0.f;
#pragma unroll
  for (int k = 0; k < 16; k++) {
    float va = Y[0][k] * E[k].x + E[k].y, vb = Y[1][k] * E[k].z + E[k].w;
    va = va > 0.f ? va : 0.f; vb = vb > 0.f ? vb : 0.f;
    float s = va + vb;
    s = s > 0.f ? s : 0.f;
    const int hh = 4 * ty + (k >> 2), ww = 4 * tx + (k & 3);
    if (hh < 19 && ww < 19) { y[((size_t)b * 441 + (hh + 1) * 21 + (ww + 1)) * C + c] = s; mx = fmaxf(mx, s); }
  }
  if (LEVEL >= 7) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0 && mx > 0.f) atomicMax(amax + b, __float_as_uint(mx));
  }
}
int main() {
  float *m, *y;
  const size_t nm = (size_t)36 * T * N, ny = (size_t)512 * 441 * C;   // y: padded NHWC of 512 boards (>= T*16*C)
  CK(hipMalloc(&m, nm * 4)); CK(hipMalloc(&y, ny * 4));
  CK(hipMemset(m, 0x3c, nm * 4)); CK(hipMemset(y, 0, ny * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch, double bytes) {
    for (int i = 0; i < 2; i++) launch();
    CK(hipEventRecord(e0));
    for (int i = 0; i < 10; i++) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    printf("%-28s %.4f ms  %.2f TB/s\n", name, ms, bytes / (ms * 1e-3) / 1e12);
  };
  const double rd = nm * 4.0, wr = ny * 4.0;
  for (int g : {256 * 8, 256 * 16, 256 * 32, 256 * 64})
    run(("P0 float4 stream, grid " + std::to_string(g)).c_str(), [&] { hipLaunchKernelGGL(p0, dim3(g), dim3(256), 0, 0, (const float4*)m, y, nm / 4); }, rd);
  run("P1 out-kernel map", [&] { hipLaunchKernelGGL((p1<false>), dim3(T * C / 256), dim3(256), 0, 0, m, y); }, rd + wr);
  run("P2 [T][36][N] layout", [&] { hipLaunchKernelGGL((p1<true>), dim3(T * C / 256), dim3(256), 0, 0, m, y); }, rd + wr);
  run("P3 dwordx2, channel pairs", [&] { hipLaunchKernelGGL(p3, dim3(T * C / 2 / 256), dim3(256), 0, 0, m, y); }, rd + wr);
  float4* ep; unsigned* amax;
  CK(hipMalloc(&ep, (size_t)361 * C * 16)); CK(hipMemset(ep, 0x3c, (size_t)361 * C * 16));
  CK(hipMalloc(&amax, 512 * 4)); CK(hipMemset(amax, 0, 512 * 4));
  const double wr2 = 512.0 * 361 * C * 4;
  run("P5 + 16 parameter vectors", [&] { hipLaunchKernelGGL((p5<5>), dim3(T * C / 256), dim3(256), 0, 0, m, ep, y, amax); }, rd + wr2);
  run("P6 + At M A and epilogue", [&] { hipLaunchKernelGGL((p5<6>), dim3(T * C / 256), dim3(256), 0, 0, m, ep, y, amax); }, rd + wr2);
  run("P7 + per-board max", [&] { hipLaunchKernelGGL((p5<7>), dim3(T * C / 256), dim3(256), 0, 0, m, ep, y, amax); }, rd + wr2);
  run("P4 two tiles per thread", [&] { hipLaunchKernelGGL(p4, dim3(T / 2 * C / 256), dim3(256), 0, 0, m, y); }, rd + wr);
  return 0;
}
