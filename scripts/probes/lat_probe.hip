// Batch-1 dual-block kernel (agogo_amd/csrc/conv_lat.hpp) on its own: 40 layers with 40 distinct 7 MB weight images (nothing stays in
// L2 between evaluations, as in the real tower), launch-to-launch time per layer and s_memtime stamps of the phases of one wave.
//   hipcc --offload-arch=gfx950 -O3 -DLAT_PROBE lat_probe.hip -o lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <type_traits>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#ifndef LAT_PROBE_X
#define LAT_PROBE_X 0
#endif
#ifndef LAT_PROBE_W
#define LAT_PROBE_W 0
#endif
struct agz_ctx { hipStream_t stream; };
namespace agz {
typedef short bf16x8_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void x3_split(float v, unsigned& h, unsigned& m, unsigned& l) {
  unsigned hu = __float_as_uint(v) & 0xffff0000u;
  float r = v - __uint_as_float(hu);
  unsigned mu = __float_as_uint(r) & 0xffff0000u;
  float r2 = r - __uint_as_float(mu);
  h = hu; m = mu; l = __float_as_uint(r2);
}
__device__ __forceinline__ unsigned x3_pack(unsigned e0, unsigned e1) { return __builtin_amdgcn_perm(e1, e0, 0x07060302u); }
__device__ __forceinline__ void h2_scales(unsigned amax_bits, float* s, float* inv) {   // (conv_h2.hpp)
  int e = (int)((amax_bits >> 23) & 0xffu);
  if (amax_bits == 0u) { *s = 1.f; *inv = 1.f; return; }
  e = e < 30 ? 30 : (e > 230 ? 230 : e);
  *s = __uint_as_float((unsigned)(267 - e) << 23);
  *inv = __uint_as_float((unsigned)(e - 13) << 23);
}
#include "../../agogo_amd/csrc/conv_lat.hpp"
}

int main(int argc, char** argv) {
  const int L = 40, C = 256, H = 19, W = 19, Hp = 21, Wp = 21, Ntot = 512;
  const size_t wbytes = (size_t)(C / 16) * 9 * 3 * Ntot * 16 * 2;
  const size_t abytes = (size_t)Hp * Wp * C * 4;
  std::vector<unsigned short*> w3(L);
  std::vector<unsigned short> hw(wbytes / 2);
  for (size_t i = 0; i < hw.size(); i++) hw[i] = (unsigned short)(0x3c00 + (rand() & 0xff));
  for (int l = 0; l < L; l++) { CK(hipMalloc(&w3[l], wbytes)); CK(hipMemcpy(w3[l], hw.data(), wbytes, hipMemcpyHostToDevice)); }
  float *x0, *x1; void* ep; long long* dbg;
  CK(hipMalloc(&x0, abytes)); CK(hipMalloc(&x1, abytes)); CK(hipMemset(x0, 0, abytes)); CK(hipMemset(x1, 0, abytes));
  std::vector<float> hep((size_t)H * W * C * 4, 0.001f);
  CK(hipMalloc(&ep, hep.size() * 4)); CK(hipMemcpy(ep, hep.data(), hep.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dbg, 64 * 8)); CK(hipMemset(dbg, 0, 64 * 8));
  agz_ctx ctx; CK(hipStreamCreate(&ctx.stream));
  agz::LatArgs a{};
  a.B = 1; a.H = H; a.W = W; a.Hp = Hp; a.Wp = Wp; a.C = C; a.Cout_p = C; a.Ntot = Ntot; a.ep = ep;
  a.groups_per_board = (H * W + agz::LAT_ROWS - 1) / agz::LAT_ROWS;
  const int pre = argc > 3 ? atoi(argv[3]) : 0;        // 1: activations arrive as bf16 pieces from the previous layer (DMA to LDS)
  unsigned short *p0, *p1;
  CK(hipMalloc(&p0, abytes / 4 * 6)); CK(hipMalloc(&p1, abytes / 4 * 6)); CK(hipMemset(p0, 0, abytes / 4 * 6)); CK(hipMemset(p1, 0, abytes / 4 * 6));
  const int gx = argc > 1 ? atoi(argv[1]) : 0, gy = argc > 2 ? atoi(argv[2]) : 8;   // grid override (fewer workgroups: latency without contention)
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto tower = [&](long long* d) {
    for (int l = 0; l < L; l++) {
      a.x = (l & 1) ? x1 : x0; a.y = (l & 1) ? x0 : x1; a.w3 = w3[l]; a.dbg = (l == L / 2) ? d : nullptr;
      a.x3 = pre ? ((l & 1) ? p1 : p0) : nullptr; a.y3 = (l & 1) ? p0 : p1;
      if (gx) hipLaunchKernelGGL((agz::conv3x3_lat_x3_kernel<8, false>), dim3(gx, gy), dim3(512), 0, ctx.stream, a);
      else agz::conv_lat_launch(&ctx, a);
    }
  };
  for (int i = 0; i < 5; i++) tower(nullptr);
  CK(hipStreamSynchronize(ctx.stream));
  const int R = 50;
  CK(hipEventRecord(e0, ctx.stream));
  for (int i = 0; i < R; i++) tower(nullptr);
  CK(hipEventRecord(e1, ctx.stream));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("tower of %d layers: %.1f us  = %.2f us per layer (launch to launch)\n", L, ms * 1e3 / R, ms * 1e3 / R / L);
  {  // the same tower as a captured graph (does the launch-to-launch gap shrink?)
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(ctx.stream, hipStreamCaptureModeGlobal));
    tower(nullptr);
    CK(hipStreamEndCapture(ctx.stream, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 5; i++) CK(hipGraphLaunch(ge, ctx.stream));
    CK(hipStreamSynchronize(ctx.stream));
    CK(hipEventRecord(e0, ctx.stream));
    for (int i = 0; i < R; i++) CK(hipGraphLaunch(ge, ctx.stream));
    CK(hipEventRecord(e1, ctx.stream));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("as a hipGraph:      %.1f us  = %.2f us per layer\n", ms * 1e3 / R, ms * 1e3 / R / L);
  }
  tower(dbg);
  CK(hipStreamSynchronize(ctx.stream));
  long long h[8]; CK(hipMemcpy(h, dbg, sizeof h, hipMemcpyDeviceToHost));
  const char* nm[7] = {"start", "activations arrived", "split -> LDS done", "weights arrived", "MFMAs issued", "reduction barrier", "stores done"};
  // s_memrealtime: constant 100 MHz
  for (int i = 1; i < 7; i++) printf("  %-22s +%6.2f us  (cum %6.2f)\n", nm[i], (h[i] - h[i - 1]) / 100.0, (h[i] - h[0]) / 100.0);
  return 0;
}
