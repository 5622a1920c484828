// Address arithmetic of the trainer's DMA forward convolutions (train.hip: k_conv_h2dma, k_conv_h2dma3), as plain constexpr functions
// that BOTH the kernels and a g++-built CPU test include (tests/cpp/conv_maps_check.cpp, run by tests/test_conv_maps_cpu.py) — the same
// arrangement as gemm_maps.hpp.
//
// Layouts: activations as two fp16 planes [padded pixel][Cin] (hi, lo; zero halo); a 32-channel chunk of a pixel is 64 bytes = four
// 16-byte units.  LDS images hold 64-byte rows; `buffer_load ... lds` writes lane-linearly (lane l of an instruction -> base + 16 l:
// sixteen rows per instruction), so the read side's bank swizzle goes on the SOURCE unit.
#pragma once
#include <cstddef>
#if defined(__HIPCC__)
#define AGZ_CMAPS_HD __host__ __device__
#else
#define AGZ_CMAPS_HD
#endif

namespace agz {
namespace cmaps {

// padded pixel index of GEMM row m (the m-th interior pixel, boards then rows then columns) of a [B][Hp][Wp] tensor, Hp = H + 2, Wp = W + 2
AGZ_CMAPS_HD constexpr size_t pix(int m, int HW, int W, int Hp, int Wp) {
  return ((size_t)(m / HW) * Hp + (size_t)((m % HW) / W) + 1) * Wp + (size_t)((m % HW) % W) + 1;
}
// LDS image of 64-byte rows: row r, 16-byte unit u at r * 64 + ((u ^ ((r >> 2) & 3)) << 4)
AGZ_CMAPS_HD constexpr unsigned lds_off(int row, int unit) { return (unsigned)(row * 64 + ((unit ^ ((row >> 2) & 3)) << 4)); }
// DMA instruction j of an image (16 rows x 4 units): lane's row, its SOURCE unit (byte offset inside the row's 64 bytes) and its
// lane-linear destination inside the image.  (row >> 2) & 3 of row 16 j + lane / 4 is (lane >> 4) & 3.
AGZ_CMAPS_HD constexpr int dma_row(int lane, int j) { return 16 * j + (lane >> 2); }
AGZ_CMAPS_HD constexpr unsigned dma_src_unit(int lane) { return (unsigned)((lane & 3) ^ ((lane >> 4) & 3)) << 4; }
AGZ_CMAPS_HD constexpr unsigned dma_dst(int lane, int j) { return (unsigned)(j * 1024 + lane * 16); }

// ---- k_conv_h2dma3: one x image per 256-pixel tile and 32-channel chunk serves all nine taps
constexpr int CD3_IMG = 372;             // rows of the image (19x19: 370 at most); a geometry that needs more keeps k_conv_h2dma
constexpr int CD3_INSTR = (CD3_IMG + 15) / 16;
// image row 0 is the padded pixel one row up and one to the left of the tile's first output pixel
AGZ_CMAPS_HD constexpr int cd3_base(size_t pix_first, int Wp) { return (int)pix_first - Wp - 1; }
// image row that tap (ky, kx) of an output pixel reads: the pixel's padded distance from the tile's first pixel + ky Wp + kx
AGZ_CMAPS_HD constexpr int cd3_row0(size_t pix_m, size_t pix_first) { return (int)pix_m - (int)pix_first; }
AGZ_CMAPS_HD constexpr int cd3_tap(int ky, int kx, int Wp) { return ky * Wp + kx; }
// rows of the image a tile whose first / last output pixels are pix_first / pix_last needs
AGZ_CMAPS_HD constexpr int cd3_rows(size_t pix_first, size_t pix_last, int Wp) { return (int)(pix_last - pix_first) + 2 * Wp + 3; }

}  // namespace cmaps
}  // namespace agz
