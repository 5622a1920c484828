// Address arithmetic of the chained Winograd GEMMs (conv_wino_h2c.hpp: wino_gemm_h2g_kernel, wino_gemm_h2p_kernel), as plain
// constexpr functions that BOTH the kernels and a g++-built CPU test include (tests/cpp/gemm_maps_check.cpp, run by
// tests/test_gemm_maps_cpu.py): the kernels compute every DMA source offset, LDS fragment address, M store address and work-list
// index through these functions, so changing one of them — or a kernel's use of one — is seen by a test without a GPU.
//
// Layouts (conv_wino_h2c.hpp): a K step of a 128-row tile of V2c is one 16 KB chunk, row r = 128 bytes = eight 16-byte units
// (0..3: hi fp16 of k 0..31, 4..7: lo).  `buffer_load ... lds` writes LDS lane-linearly (lane l of an instruction -> base + 16 l),
// so the bank swizzle of the LDS image goes on the SOURCE offset.
#pragma once
#include <cstddef>
#if defined(__HIPCC__)
#define AGZ_MAPS_HD __host__ __device__
#else
#define AGZ_MAPS_HD
#endif

namespace agz {
namespace maps {

// LDS image of a staged chunk: row r, 16-byte unit q at r * 128 + ((q ^ ((r >> 1) & 7)) << 4)
AGZ_MAPS_HD constexpr unsigned h2c_img(int row, int unit) { return (unsigned)(row * 128 + ((unit ^ ((row >> 1) & 7)) << 4)); }

// DMA instruction j of a wave (8 rows x 8 units = 1 KB, rows 8 j .. 8 j + 7 of the wave's part): lane's SOURCE byte offset inside
// the wave's part of the chunk, and its (lane-linear) LDS destination.  (r >> 1) & 7 of row 8 j + lane / 8 (+ the wave's first row,
// a multiple of 16) is (4 j + lane / 16) & 7.
AGZ_MAPS_HD constexpr unsigned h2c_dma_src(int lane, int j) {
  return (unsigned)(j * 1024 + (lane >> 3) * 128 + (((lane & 7) ^ ((4 * j + (lane >> 4)) & 7)) << 4));
}
AGZ_MAPS_HD constexpr unsigned h2c_dma_dst(int lane, int j) { return (unsigned)(j * 1024 + lane * 16); }
// the four waves of a workgroup split a stage of stage_rows rows evenly: wave w owns rows w * stage_rows / 4 .. (bytes, source = image)
AGZ_MAPS_HD constexpr unsigned h2c_wave_part(int wid, int stage_rows) { return (unsigned)(wid * (stage_rows / 4) * 128); }
AGZ_MAPS_HD constexpr int h2c_wave_instrs(int stage_rows) { return stage_rows / 4 / 8; }

// MFMA 32x32x16 operand fragment of image rows row0 .. row0 + 31 (row0 a multiple of 32): lane l reads row row0 + (l & 31), the 16-byte
// unit piece * 4 + 2 ks + (l >> 5) — piece 0 = hi, 1 = lo; k = 16 ks + 8 (l >> 5) .. + 7 of the 32-channel step
AGZ_MAPS_HD constexpr unsigned h2c_frag(int row0, int lane, int piece, int ks) {
  return (unsigned)((row0 + (lane & 31)) * 128 + (((piece * 4 + 2 * ks + (lane >> 5)) ^ ((lane >> 1) & 7)) << 4));
}

// Accumulator register r (0..15) of a 32x32 MFMA result holds tile row mfma_row(r) in lanes 0..31 and mfma_row(r) + 4 in lanes 32..63
AGZ_MAPS_HD constexpr int mfma_row(int r) { return (r & 3) + 8 * (r >> 2); }

// Mc[T/128][pos][Ntot/64 slices][128 rows][64 cols] fp32: float index of (m_tile, pos, slice, row, col)
AGZ_MAPS_HD constexpr size_t mc_index(int m_tile, int npos, int pos, int n_slices, int slice, int row, int col) {
  return ((((size_t)m_tile * npos + pos) * (size_t)n_slices + (size_t)slice) * 128 + (size_t)row) * 64 + (size_t)col;
}

// ---- persistent kernel (wino_gemm_h2p_kernel): 8 KB stages of 64 rows, ring of 16, DMA 10 stages ahead
constexpr int H2P_D = 10, H2P_R = 16, H2P_NK = 8;
// byte offset in V2c of K step 0 of half tile hm (rows 64 (hm & 1) .. of m-tile hm >> 1) of position pos
AGZ_MAPS_HD constexpr unsigned h2p_v_base(int pos, int hm, int npos) {
  return (unsigned)(((hm >> 1) * npos + pos) * (H2P_NK * 16384) + (hm & 1) * 8192);
}
// tile parity PAR (ring half), K step kk: ring slot of the step's own stage, of the next step's stage, of the stage DMA'd at this step
AGZ_MAPS_HD constexpr int h2p_slot(int par, int kk) { return (par * H2P_NK + kk) % H2P_R; }
AGZ_MAPS_HD constexpr int h2p_slot_next(int par, int kk) { return (par * H2P_NK + kk + 1) % H2P_R; }
AGZ_MAPS_HD constexpr int h2p_slot_ahead(int par, int kk) { return (par * H2P_NK + kk + H2P_D) % H2P_R; }
AGZ_MAPS_HD constexpr int h2p_kk_ahead(int kk) { return (kk + H2P_D) % H2P_NK; }
AGZ_MAPS_HD constexpr int h2p_tiles_ahead(int kk) { return (kk + H2P_D) / H2P_NK; }      // 1 or 2
// counted waits (s_waitcnt vmcnt): operations a wave issues per K step = 2 DMA + 4 M stores; "stage g + 1 has landed" when at most
// this many younger operations are outstanding — steady state: the 4 stores of the issuing step + (D - 2) whole steps; while no
// stores are in flight yet (first pair of tiles): the DMAs of D - 2 stages
AGZ_MAPS_HD constexpr int h2p_wait_steady() { return 4 + (H2P_D - 2) * 6; }
AGZ_MAPS_HD constexpr int h2p_wait_early() { return (H2P_D - 2) * 2; }
AGZ_MAPS_HD constexpr bool h2p_early(int t) { return t < 2; }                        // tiles whose waits use the early count
// work list: workgroup id -> (team, slab); team t of nteams walks units [u0, u0 + nT) of U2 = npos x n_mtiles PAIRS of half tiles
struct H2pTeam { int team, slab, nteams; bool idle; };
AGZ_MAPS_HD constexpr H2pTeam h2p_team(int block, int grid, int n_slabs) {
  const int xcd = block & 7, slot = block >> 3, tpx = (grid >> 3) / n_slabs;
  return H2pTeam{xcd * tpx + (tpx ? slot / n_slabs : 0), tpx ? slot % n_slabs : 0, 8 * tpx, slot >= tpx * n_slabs};
}
AGZ_MAPS_HD constexpr int h2p_u0(int team, int nteams, int U2) { return 2 * (int)((long)team * U2 / nteams); }

}  // namespace maps
}  // namespace agz
