// micro-probe: achievable v_mfma_f32_32x32x2_f32 issue rate in the conv kernel's inner-loop shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int VAR>
__global__ __launch_bounds__(256) void probe(const float* in, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[16384 + (VAR >= 10 ? 10240 : 0)];
  int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 16384; i += 256) lds[i] = in[i & 1023];
  __syncthreads();
  f32x16 acc[2][2];
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;
  const float* base = lds + (lane & 31) * 32 + (lane >> 5) * 4;
  float4 a0 = *(const float4*)(base), a1 = *(const float4*)(base + 1024), b0 = *(const float4*)(base + 2048), b1 = *(const float4*)(base + 3072);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int ks = 0; ks < 4; ks++) {
      float4 na0 = a0, na1 = a1, nb0 = b0, nb1 = b1;
      if (VAR % 10 >= 1) {  // LDS operand reads for the next k-step, issued before this step's MFMAs
        const float* p = base + ((it * 4 + ks) & 3) * 4096 + ks * 8;
        na0 = *(const float4*)(p); na1 = *(const float4*)(p + 1024); nb0 = *(const float4*)(p + 2048); nb1 = *(const float4*)(p + 3072);
        __builtin_amdgcn_sched_barrier(0);
      }
#define M4(X) \
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.X, b0.X, acc[0][0], 0, 0, 0); \
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.X, b1.X, acc[0][1], 0, 0, 0); \
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.X, b0.X, acc[1][0], 0, 0, 0); \
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.X, b1.X, acc[1][1], 0, 0, 0);
      M4(x) M4(y) M4(z) M4(w)
      __builtin_amdgcn_sched_barrier(0);
      a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
    }
  }
  float s = 0;
  for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 16; r++) s += acc[i][j][r];
  out[blockIdx.x * 256 + tid] = s;
}
template <int VAR>
void run(const char* name, int blocks_per_cu) {
  int iters = 2000;
  float *in, *out;
  hipMalloc(&in, 4096); hipMalloc(&out, 4 * 256 * 256 * 4);
  hipMemset(in, 0, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int grid = 256 * blocks_per_cu;
  probe<VAR><<<grid, 256>>>(in, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<VAR><<<grid, 256>>>(in, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)grid * 4 * iters * 64.0 * 4096.0;
  printf("%-28s blocks/CU=%d  %.3f ms  %.1f TF  (%.1f%% of 157.3)\n", name, blocks_per_cu, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}
template <int VAR>
void run_turnover(const char* name, int nblocks, int iters) {
  float *in, *out;
  hipMalloc(&in, 4096); hipMalloc(&out, (size_t)nblocks * 256 * 4);
  hipMemset(in, 0, 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<VAR><<<nblocks, 256>>>(in, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<VAR><<<nblocks, 256>>>(in, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)nblocks * 4 * iters * 64.0 * 4096.0;
  printf("%-28s blocks=%d iters=%d  %.3f ms  %.1f TF  (%.1f%% of 157.3)\n", name, nblocks, iters, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.3 * 100);
}
int main() {
  run_turnover<1>("turnover 64KB-lds", 5776, 72);
  run_turnover<1>("turnover 64KB-lds", 5632, 72);   // 11 exact rounds of 512
  run_turnover<1>("turnover 64KB-lds", 512, 72 * 11);
  run_turnover<1>("turnover 64KB-lds", 512 * 44, 18);
  run<0>("mfma only", 1); run<0>("mfma only", 2);
  run<1>("mfma + ds_read prefetch", 1); run<1>("mfma + ds_read prefetch", 2);
  return 0;
}
