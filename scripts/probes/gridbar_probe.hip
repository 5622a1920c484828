// Grid-wide barrier cost on MI355X (256 workgroups of 512 threads, one per CU, cooperative launch): what a persistent tower kernel
// would pay per layer instead of a kernel boundary.  Each phase: every workgroup writes a 1.5 KB slice, release + atomic arrive,
// spin (acquire), then reads a slice another XCD wrote in this phase and checks it.
//   hipcc --offload-arch=gfx950 -O3 gridbar_probe.hip -o gridbar_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(512) void k(unsigned* ctr, float* buf, int phases, unsigned* bad, long long* t) {
  const int nwg = gridDim.x, wg = blockIdx.x, tid = threadIdx.x;
  unsigned errs = 0;
  long long t0 = 0;
  for (int ph = 0; ph < phases; ph++) {
    if (ph == 8 && wg == 0 && tid == 0) t0 = __builtin_amdgcn_s_memrealtime();
    {
      float* dst = &buf[(size_t)(ph & 1) * nwg * 512 + wg * 512 + tid];
      const float val = (float)(ph * 1000 + wg);
      // write-through store (sc0 sc1: visible at the memory side once acknowledged), no L2 write-back / invalidate instructions anywhere
      asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" :: "v"(dst), "v"(val) : "memory");
    }
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(ph + 1) * nwg;
      int spins = 0;
      while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(0);
        if (++spins > 2000000) { atomicAdd(bad, 1000000u); break; }                         // (never hang the box)
      }
    }
    __syncthreads();
    const int other = (wg + 3) % nwg;                                                       // a workgroup of another XCD
    float got;
    {
      const float* src = &buf[(size_t)(ph & 1) * nwg * 512 + other * 512 + tid];
      asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(got) : "v"(src) : "memory");
    }
    if (got != (float)(ph * 1000 + other)) errs++;
  }
  if (wg == 0 && tid == 0) { t[0] = t0; t[1] = __builtin_amdgcn_s_memrealtime(); }
  if (errs) atomicAdd(bad, errs);
}

int main() {
  unsigned *ctr, *bad; float* buf; long long* t;
  const int nwg = 256, phases = 1008;
  CK(hipMalloc(&ctr, 4)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&buf, (size_t)2 * nwg * 512 * 4)); CK(hipMalloc(&t, 16));
  CK(hipMemset(ctr, 0, 4)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(buf, 0, (size_t)2 * nwg * 512 * 4));
  void* args[] = {&ctr, &buf, (void*)&phases, &bad, &t};
  CK(hipLaunchCooperativeKernel((void*)k, dim3(nwg), dim3(512), args, 0, 0));
  CK(hipDeviceSynchronize());
  unsigned hb; long long ht[2];
  CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ht, t, 16, hipMemcpyDeviceToHost));
  printf("grid barrier + 1.5 KB exchange, %d workgroups: %.2f us per phase, %u stale/failed reads\n", nwg, (ht[1] - ht[0]) / 100.0 / (phases - 8), hb);
  return 0;
}
