// L2 -> CU operand-stream probe (MI355X): how many bytes per clock can one CU pull from an L2-resident (or HBM-resident) region,
// by which instruction, at which occupancy?   hipcc --offload-arch=gfx950 -O3 l2_probe.hip -o l2_probe
// Every workgroup streams its own region of REGION bytes ITERS times (regions of an XCD's workgroups together fit / do not fit L2).
//   mode 0  LDS-DMA   buffer_load_dwordx4 ... lds, 1 KB contiguous per wave instruction, DEPTH instructions in flight per wave
//   mode 1  registers global_load_dwordx4 (16 B per lane, 1 KB contiguous per instruction), DEPTH in flight per wave, summed
//   mode 2  registers + ds_write_b128 into an LDS ring (the register-staged operand path)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE, int DEPTH, int THREADS>
__global__ __launch_bounds__(THREADS) void probe(const char* __restrict__ src, float* __restrict__ out, unsigned region, int iters, unsigned wg_stride) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NW = THREADS / 64;
  const char* base = src + (size_t)blockIdx.x * wg_stride;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, 0x7fffffff, 0x00020000);
  const unsigned voff = (unsigned)w * 1024u + (unsigned)lane * 16u;
  const unsigned step = (unsigned)NW * 1024u;            // bytes the workgroup takes per "row" of instructions
  const unsigned nrow = region / step;
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; it++) {
    for (unsigned row = 0; row < nrow; row += DEPTH) {
      if (MODE == 0) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(lds + (d % 8) * (NW * 1024) % 65536 + w * 1024), 16, voff, (row + d) * step, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      } else {
        u32x4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) v[d] = __builtin_amdgcn_raw_buffer_load_b128(r, voff, (row + d) * step, 0);
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
          if (MODE == 2) *reinterpret_cast<u32x4*>(lds + ((d % 4) * (NW * 1024)) % 65536 + w * 1024 + lane * 16) = v[d];
          else acc += v[d];
        }
      }
    }
  }
  if (MODE == 2) { __syncthreads(); acc += *reinterpret_cast<u32x4*>(lds + lane * 16); }
  if (MODE == 0) { __syncthreads(); acc += *reinterpret_cast<u32x4*>(lds + lane * 16); }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678u) out[0] = 1.f;
}

template <int MODE, int DEPTH, int THREADS>
static void run(const char* name, const char* d_src, float* d_out, unsigned region, int iters, unsigned wg_stride, int nwg, double clk_ghz) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe<MODE, DEPTH, THREADS>), dim3(nwg), dim3(THREADS), 0, 0, d_src, d_out, region, 2, wg_stride);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe<MODE, DEPTH, THREADS>), dim3(nwg), dim3(THREADS), 0, 0, d_src, d_out, region, iters, wg_stride);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)nwg * region * iters;
  printf("%-44s region %7u B x %4d wgs x %3d it  %8.3f ms  %7.2f TB/s  %6.1f B/clk/CU (256 CUs @ %.1f GHz)\n", name, region, nwg, iters, ms,
         bytes / ms * 1e-9, bytes / ms * 1e-6 / 256.0 / clk_ghz, clk_ghz);
}

int main() {
  const size_t total = (size_t)1 << 31;   // 2 GiB source
  char* d_src; float* d_out;
  CK(hipMalloc(&d_src, total)); CK(hipMalloc(&d_out, 64));
  CK(hipMemset(d_src, 1, total));
  const double clk = 2.1;
  // L2-resident: 256 workgroups x 64 KB (2 MB per XCD), 512 x 32 KB
  run<0, 4, 512>("DMA  depth 4, 512 thr, 1 wg/CU, L2", d_src, d_out, 65536, 2000, 65536, 256, clk);
  run<0, 8, 512>("DMA  depth 8, 512 thr, 1 wg/CU, L2", d_src, d_out, 65536, 2000, 65536, 256, clk);
  run<0, 8, 256>("DMA  depth 8, 256 thr, 2 wg/CU, L2", d_src, d_out, 32768, 2000, 32768, 512, clk);
  run<0, 8, 256>("DMA  depth 8, 256 thr, 4 wg/CU, L2", d_src, d_out, 16384, 2000, 16384, 1024, clk);
  run<1, 4, 512>("REG  depth 4, 512 thr, 1 wg/CU, L2", d_src, d_out, 65536, 2000, 65536, 256, clk);
  run<1, 8, 512>("REG  depth 8, 512 thr, 1 wg/CU, L2", d_src, d_out, 65536, 2000, 65536, 256, clk);
  run<1, 8, 256>("REG  depth 8, 256 thr, 2 wg/CU, L2", d_src, d_out, 32768, 2000, 32768, 512, clk);
  run<1, 8, 256>("REG  depth 8, 256 thr, 4 wg/CU, L2", d_src, d_out, 16384, 2000, 16384, 1024, clk);
  run<1, 8, 256>("REG  depth 8, 256 thr, 8 wg/CU, L2", d_src, d_out, 8192, 2000, 8192, 2048, clk);
  run<2, 8, 512>("REG+ds_write depth 8, 512 thr, 1 wg/CU, L2", d_src, d_out, 65536, 2000, 65536, 256, clk);
  run<2, 8, 256>("REG+ds_write depth 8, 256 thr, 2 wg/CU, L2", d_src, d_out, 32768, 2000, 32768, 512, clk);
  // shared region: all workgroups of the chip read the SAME 64 KB (broadcast-like: weights)
  run<0, 8, 512>("DMA  depth 8, 512 thr, same 64 KB for all", d_src, d_out, 65536, 2000, 0, 256, clk);
  run<1, 8, 512>("REG  depth 8, 512 thr, same 64 KB for all", d_src, d_out, 65536, 2000, 0, 256, clk);
  // HBM-resident: 256 workgroups x 4 MB (1 GiB), once
  run<0, 8, 512>("DMA  depth 8, 512 thr, 1 wg/CU, HBM", d_src, d_out, 4u << 20, 2, 4u << 20, 256, clk);
  run<1, 8, 512>("REG  depth 8, 512 thr, 1 wg/CU, HBM", d_src, d_out, 4u << 20, 2, 4u << 20, 256, clk);
  run<1, 8, 256>("REG  depth 8, 256 thr, 4 wg/CU, HBM", d_src, d_out, 1u << 20, 2, 1u << 20, 1024, clk);
  // MALL-resident: 256 workgroups x 512 KB (128 MB), 8 passes
  run<0, 8, 512>("DMA  depth 8, 512 thr, 1 wg/CU, 128 MB", d_src, d_out, 512u << 10, 16, 512u << 10, 256, clk);
  run<1, 8, 512>("REG  depth 8, 512 thr, 1 wg/CU, 128 MB", d_src, d_out, 512u << 10, 16, 512u << 10, 256, clk);
  return 0;
}
