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
int main() {
  float *m, *y;
  const size_t nm = (size_t)36 * T * N, ny = (size_t)T * 16 * C;
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
  run("P4 two tiles per thread", [&] { hipLaunchKernelGGL(p4, dim3(T / 2 * C / 256), dim3(256), 0, 0, m, y); }, rd + wr);
  return 0;
}
