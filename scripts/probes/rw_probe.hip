// HBM ceiling for MIXED read / write streams on MI355X: what can a kernel that reads R bytes and writes W bytes (both streamed once,
// nothing re-used) reach?  The Winograd GEMM moves 0.44 GB in + 0.82 GB out per launch, the transforms 0.19 + 0.41 and 0.82 + 0.19.
//   hipcc --offload-arch=gfx950 -O3 rw_probe.hip -o rw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// every thread: RN float4 loads and WN float4 stores per iteration, grid-stride over 16-byte words; runs of 1 KB per wave instruction
template <int RN, int WN>
__global__ __launch_bounds__(256) void rw(const float4* __restrict__ src, float4* __restrict__ dst, size_t iters) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nth = (size_t)gridDim.x * 256;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t it = 0; it < iters; it++) {
    float4 v[RN > 0 ? RN : 1];
#pragma unroll
    for (int k = 0; k < RN; k++) v[k] = src[(it * RN + k) * nth + tid];
#pragma unroll
    for (int k = 0; k < RN; k++) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
#pragma unroll
    for (int k = 0; k < WN; k++) dst[(it * WN + k) * nth + tid] = make_float4(acc.x + k, acc.y, acc.z, acc.w);
  }
  if (WN == 0 && acc.x == 1.2345f) dst[tid] = acc;
}

template <int RN, int WN>
static void run(const char* name, const float4* src, float4* dst, size_t unit_bytes, int grid) {
  const size_t nth = (size_t)grid * 256;
  const size_t iters = unit_bytes / (nth * 16);          // per-iteration: RN reads + WN writes of nth float4
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((rw<RN, WN>), dim3(grid), dim3(256), 0, 0, src, dst, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < 5; i++) hipLaunchKernelGGL((rw<RN, WN>), dim3(grid), dim3(256), 0, 0, src, dst, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  const double rb = (double)iters * RN * nth * 16, wb = (double)iters * WN * nth * 16;
  printf("%-28s grid %5d: read %.3f GB + write %.3f GB in %.3f ms = %.2f TB/s\n", name, grid, rb / 1e9, wb / 1e9, ms, (rb + wb) / ms / 1e9);
}

int main() {
  // round 4: streams of >= 2 GB per launch — the 0.4 GB / 0.08 ms launches of round 3 read 5.3 TB/s where
  // stream_probe (round 2) and the guide read 6.3-6.5: at that size the launch ramp and the tail are a fifth of the kernel
  const size_t cap = (size_t)8 << 30;
  float4 *src, *dst;
  CK(hipMalloc(&src, cap)); CK(hipMalloc(&dst, cap));
  CK(hipMemset(src, 0, cap)); CK(hipMemset(dst, 0, cap));
  const size_t unit = (size_t)2 << 30;                    // bytes per "1" of the ratio
  for (int grid : {2048, 8192}) {
    run<4, 0>("read only (4:0)", src, dst, unit / 4, grid);          // 4 x unit/4... = unit read per launch x4 (RN loads per iteration)
    run<0, 4>("write only (0:4)", src, dst, unit / 4, grid);
    run<2, 2>("copy (1:1)", src, dst, unit / 2, grid);
    run<1, 2>("GEMM mix (1 : 2)", src, dst, unit, grid);
    run<4, 1>("output transform mix (4 : 1)", src, dst, unit / 2, grid);
    run<1, 2>("input transform mix (1 : 2)", src, dst, unit / 2, grid);
    run<2, 1>("out->in (chained) mix (2 : 1)", src, dst, unit, grid);
  }
  return 0;
}
