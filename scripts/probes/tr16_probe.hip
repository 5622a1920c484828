// Probe: semantics of ds_read_b64_tr_b16 (gfx950).  Every lane supplies its own 8-byte LDS address; the probe prints, for every
// lane and element, WHICH (supplying lane, element) the value came from.  Model to confirm: within a 16-lane group,
// result[l][j] = piece[j * 4 + (l & 15) / 4][(l & 15) % 4], piece[i] = the 4 halves at the address lane i of the group supplied.
// build: hipcc --offload-arch=gfx950 -O2 tr16_probe.hip -o tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((vector_size(8)));
__global__ void k(unsigned short* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 4 * 4];
  const int l = threadIdx.x;
  // lane i's piece lives at a scattered place: slot = (i * 7) % 64 (mode 0) or i (mode 1); its halves carry (i << 2 | e)
  const int slot = mode ? l : (l * 7) % 64;
  for (int e = 0; e < 4; e++) lds[slot * 4 + e] = (unsigned short)(l << 2 | e);
  __syncthreads();
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + slot * 4));
  for (int j = 0; j < 4; j++) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  int bad = 0;
  for (int mode = 0; mode < 2; mode++) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l++)
      for (int j = 0; j < 4; j++) {
        const int src_lane = h[l * 4 + j] >> 2, src_e = h[l * 4 + j] & 3;
        const int want_lane = (l & ~15) + j * 4 + (l & 15) / 4, want_e = (l & 15) % 4;
        if (src_lane != want_lane || src_e != want_e) { if (bad < 8) printf("mode %d lane %d elem %d: from lane %d elem %d (model: lane %d elem %d)\n", mode, l, j, src_lane, src_e, want_lane, want_e); bad++; }
      }
  }
  printf("tr16 model %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
  return bad != 0;
}
