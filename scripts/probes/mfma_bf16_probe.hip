// Ceiling probe: v_mfma_f32_32x32x16_bf16 issue rate with W waves/SIMD and nothing else (power-limited clock shows up
// as TF/s below the 2.4 GHz datasheet figure).  hipcc --offload-arch=gfx950 -O3 mfma_bf16_probe.hip -o mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, short seed) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
  // six different operand pairs, rotated like a real K loop rotates fragments.  seed 0: zeros; seed 1: random bf16 in
  // [-2,2) (random sign/exponent-low/mantissa bits = realistic toggling); other: a smooth small-integer pattern
  bf16x8_t a[6], b[6];
  unsigned x = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x;
  for (int u = 0; u < 6; u++)
    for (int e = 0; e < 8; e++) {
      x = x * 1664525u + 1013904223u; unsigned ra = x >> 16;
      x = x * 1664525u + 1013904223u; unsigned rb = x >> 16;
      if (seed == 0) { a[u][e] = 0; b[u][e] = 0; }
      else if (seed == 1) { a[u][e] = (short)((ra & 0x807f) | (0x7e + (ra >> 7 & 1) * 1) << 7); b[u][e] = (short)((rb & 0x807f) | (0x7e + (rb >> 7 & 1)) << 7); }
      else { a[u][e] = (short)(seed + threadIdx.x + e); b[u][e] = (short)(seed * 3 + threadIdx.x * 7 + e); }
    }
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 6; u++)
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[(u + i) % 6], acc[i], 0, 0, 0);
    // keep the accumulators bounded so they do not saturate to inf (which would stop toggling)
    if ((it & 63) == 63)
      for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) acc[i][r] *= 1e-3f;
  }
  float s = 0;
  for (int i = 0; i < NACC; i++) for (int r = 0; r < 16; r++) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char** argv) {
  int blocks_per_cu = argc > 1 ? atoi(argv[1]) : 3;
  short seed = argc > 2 ? (short)atoi(argv[2]) : 0x3f80;   // 0: zero operands (lower power)
  int iters = 20000, nblk = 256 * blocks_per_cu;
  float* out; hipMalloc(&out, nblk * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<4>), dim3(nblk), dim3(256), 0, 0, out, iters, seed);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)nblk * 4 * iters * 24 * (32.0 * 32 * 16 * 2);
    printf("blocks/CU %d seed %d: %.3f ms  %.1f TFLOP/s bf16  (%.1f fp32-equivalent via 6 products)\n", blocks_per_cu, seed, ms, flops / ms / 1e9, flops / ms / 1e9 / 6);
  }
  return 0;
}
