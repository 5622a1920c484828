// Probe (round 6): are the float operations select_child (engine.hip) relies on CORRECTLY ROUNDED on this toolchain?
// __fsqrt_rn is __ocml_native_sqrt_f32 in this ROCm's __clang_hip_math.h (1 ulp), found through a PUCT tie the narrow-tree fuzz produced:
// sqrt(300.f) = 0x418a9066 on the device, 0x418a9067 correctly rounded.  Counts mismatches against the host's IEEE results for
// (a) sqrt of every integer 1 .. 2^24 by four device formulations, (b) num / den over sqrt values and small integer denominators.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
__device__ __forceinline__ float sqrt_cr(float x) {   // correctly rounded by construction: native estimate, exact bracketing and midpoint test in double
  float r = __builtin_amdgcn_sqrtf(x);
  while ((double)r * (double)r > (double)x) r = __uint_as_float(__float_as_uint(r) - 1u);
  float rn = __uint_as_float(__float_as_uint(r) + 1u);
  while ((double)rn * (double)rn <= (double)x) { r = rn; rn = __uint_as_float(__float_as_uint(r) + 1u); }
  const double mid = 0.5 * ((double)r + (double)rn);
  return (double)x > mid * mid ? rn : r;
}
__global__ void k_sqrt(const float* ref, unsigned n, unsigned* bad) {
  unsigned i = blockIdx.x * 256 + threadIdx.x + 1;
  if (i > n) return;
  const float x = (float)i, want = ref[i];
  if (__float_as_uint(__fsqrt_rn(x)) != __float_as_uint(want)) atomicAdd(&bad[0], 1u);
  if (__float_as_uint(sqrtf(x)) != __float_as_uint(want)) atomicAdd(&bad[1], 1u);
  if (__float_as_uint((float)sqrt((double)x)) != __float_as_uint(want)) atomicAdd(&bad[2], 1u);
  if (__float_as_uint(sqrt_cr(x)) != __float_as_uint(want)) atomicAdd(&bad[3], 1u);
}
__global__ void k_div(const float* num, const float* den, const float* ref, unsigned n, unsigned* bad) {
  unsigned i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  if (__float_as_uint(__fdiv_rn(num[i], den[i])) != __float_as_uint(ref[i])) atomicAdd(&bad[4], 1u);
  if (__float_as_uint(num[i] / den[i]) != __float_as_uint(ref[i])) atomicAdd(&bad[5], 1u);
  if (__float_as_uint((float)((double)num[i] / (double)den[i])) != __float_as_uint(ref[i])) atomicAdd(&bad[6], 1u);
}
int main() {
  const unsigned N = 1u << 24;
  std::vector<float> ref(N + 1);
  for (unsigned i = 1; i <= N; i++) { volatile float x = (float)i; ref[i] = std::sqrt(x); }
  float* dref; unsigned* dbad;
  hipMalloc(&dref, (N + 1) * 4); hipMalloc(&dbad, 32); hipMemset(dbad, 0, 32);
  hipMemcpy(dref, ref.data(), (N + 1) * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_sqrt, dim3((N + 255) / 256), dim3(256), 0, 0, dref, N, dbad);
  const unsigned M = 1u << 22;
  std::vector<float> num(M), den(M), q(M);
  unsigned long long s = 88172645463325252ull;
  for (unsigned i = 0; i < M; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    volatile float a = std::sqrt((float)(1 + (s % 1000000))), b = (float)(1 + ((s >> 32) % 100000));
    num[i] = a; den[i] = b; volatile float c = a / b; q[i] = c;
  }
  float *dn, *dd, *dq;
  hipMalloc(&dn, M * 4); hipMalloc(&dd, M * 4); hipMalloc(&dq, M * 4);
  hipMemcpy(dn, num.data(), M * 4, hipMemcpyHostToDevice); hipMemcpy(dd, den.data(), M * 4, hipMemcpyHostToDevice); hipMemcpy(dq, q.data(), M * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k_div, dim3((M + 255) / 256), dim3(256), 0, 0, dn, dd, dq, M, dbad);
  unsigned bad[8];
  hipMemcpy(bad, dbad, 32, hipMemcpyDeviceToHost);
  printf("sqrt of 1..2^24, mismatches vs IEEE: __fsqrt_rn %u, sqrtf %u, (float)sqrt((double)x) %u, sqrt_cr %u\n", bad[0], bad[1], bad[2], bad[3]);
  printf("num/den over %u pairs, mismatches vs IEEE: __fdiv_rn %u, operator/ %u, via double %u\n", M, bad[4], bad[5], bad[6]);
  return 0;
}
