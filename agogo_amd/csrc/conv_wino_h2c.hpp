// Chained form of the Winograd fp16x2 tower (AGZ_COMPUTE_WINO_H2, reference shape dualnet/ermahagerdmonards.go:60-73: `share` =
// two convolutions of the same input, add, ReLU).
//
// conv_wino_h2.hpp runs a block as  x -> [in] -> V2 -> [GEMM] -> M -> [out] -> y  and the next block reads y again: of the
// 2.87 GB a 512-board block moves, 0.38 GB is y written and read back.  Here the output transform of block l and the input
// transform of block l+1 are ONE kernel (wino_oi_h2c_kernel): a workgroup owns 16 tile rows (a 19x19 board = 4x4 tiles of
// F(5x5,3x3)) x one 32-channel slice, reads its slice of M(l), applies At . A, the block epilogue, keeps the 32-channel board in
// LDS (46 KB), and writes Bt . B of it as the split fp16 operand V2(l+1).  y never reaches HBM between blocks: 2.49 GB per block.
//
// Range words.  V2(l+1) needs the board's power-of-two range BEFORE the board's maximum over all channel slices exists (the
// slices are different workgroups), so the range comes from a bound proven at commit time:
//     max |y_l * t_next|  <=  g1_l * max |x_l|  +  g0_l          (g1 = max_{p,c} t_next (|sa| L1(w_a,c) + |sb| L1(w_b,c)),  g0 = max (ta+ + tb+))
// with max|x_l| the EXACT maximum of the block's input (per-(tile, slice) words left by the producing kernel, reduced by every
// consumer itself — ping-pong arrays).  The bound is one layer deep (no compounding) and costs log2(bound / true max) bits of
// the 2^-38 absolute piece error (conv_wino_h2.hpp header): ~5 bits on Glorot weights, invisible next to fp32's 2^-24.
//
// Layouts (all 16-byte runs of a wave are contiguous KBs):
//   V2c[T/128][pos][C/32][128 rows][hi 32 fp16 | lo 32 fp16]      a GEMM K step of a 128-row tile = one 16 KB chunk
//   U2c[pos][C/32][Ntot/256][256 cols][hi | lo]                    a K step of a 256-column tile = one 32 KB chunk
//   Mc [T/128][pos][C/32 slices][128 rows][64 cols] fp32           column 2j + br of slice s = branch br of channel 32 s + j
// LDS image of a staged chunk: row r, 16-byte slot q (0..3 hi, 4..7 lo) at r * 128 + ((q ^ ((r >> 1) & 7)) << 4): conflict-free
// for the 128-byte row writes and for the MFMA fragment reads (ds_read_b128: even and odd rows are the two halves of the
// 64-bank period, and the 8 even / 8 odd rows of a 16-lane group get 8 distinct slots).
#pragma once
// (included by net.hip INSIDE namespace agz, after conv_wino_h2.hpp)

// (every offset of the two DMA GEMMs below comes from gemm_maps.hpp — plain constexpr functions a CPU test checks, tests/test_gemm_maps_cpu.py)
__device__ __forceinline__ unsigned h2c_img(int row, int slot) { return maps::h2c_img(row, slot); }

// ---- the transform-domain GEMMs on the chained layouts: 128 x 256 tile, A (HBM) fetched PFA K steps ahead into rotating register
// sets, B (L2) one step ahead — wino_gemm_h2d_kernel's pipeline; every global access of a wave is one contiguous KB.
template <int NK, int PFA>
__global__ __launch_bounds__(256, 2) void wino_gemm_h2c_kernel(WinoH2Args h) {
  const WinoArgs& a = h.w;
  constexpr int SA = 128 * 128, SB = 256 * 128;   // 16 KB + 32 KB
  __shared__ __attribute__((aligned(16))) unsigned char lds[SA + SB];

  const int n_nt = a.Ntot >> 8;
  const int per_pos = a.n_mtiles * n_nt;
  const int nblk = h.npos * per_pos;
  const int id = blockIdx.x;
  int q = nblk >> 3, rr = nblk & 7, xcd = id & 7, slot = id >> 3;
  int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + slot;
  const int pos = tile / per_pos;
  const int rem = tile - pos * per_pos;
  const int m_tile = rem / n_nt, n_tile = rem - m_tile * n_nt;

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int wm = wid >> 1, wn = wid & 1;
  // staging: thread t moves the 16-byte unit t + 256 j of a chunk (row t/8 + 32 j, slot t%8)
  const unsigned st_lds = h2c_img(tid >> 3, tid & 7);          // + j * 4096: (row >> 1) & 7 does not depend on j
  const char* abase = reinterpret_cast<const char*>(a.V) + ((size_t)m_tile * h.npos + pos) * (size_t)NK * SA + (size_t)tid * 16;
  const char* bbase = reinterpret_cast<const char*>(h.U2c) + (((size_t)pos * NK) * n_nt + n_tile) * (size_t)SB + (size_t)tid * 16;
  const size_t bstep = (size_t)n_nt * SB;

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

  const int kh = lane >> 5, sw = (lane >> 1) & 7;
  const unsigned fa = (unsigned)((wm * 64 + (lane & 31)) * 128);          // + i * 4096
  const unsigned fb = (unsigned)(SA + (wn * 128 + (lane & 31)) * 128);    // + j * 4096

  u32x4_t xa[PFA][4], xb[8];
  auto load_a = [&](int set, int kk) {
#pragma unroll
    for (int j = 0; j < 4; j++) xa[set][j] = *reinterpret_cast<const u32x4_t*>(abase + (size_t)kk * SA + j * 4096);
  };
  auto load_b = [&](int kk) {
#pragma unroll
    for (int j = 0; j < 8; j++) xb[j] = *reinterpret_cast<const u32x4_t*>(bbase + (size_t)kk * bstep + j * 4096);
  };
  load_b(0);
#pragma unroll
  for (int p = 0; p < PFA; p++) load_a(p, p < NK ? p : NK - 1);
#pragma unroll
  for (int it = 0; it < NK; it++) {
    const int set = it % PFA;
#pragma unroll
    for (int j = 0; j < 4; j++) *reinterpret_cast<u32x4_t*>(lds + st_lds + j * 4096) = xa[set][j];
#pragma unroll
    for (int j = 0; j < 8; j++) *reinterpret_cast<u32x4_t*>(lds + SA + st_lds + j * 4096) = xb[j];
    __syncthreads();
    if (it + 1 < NK) load_b(it + 1);            // B first (vmcnt retires in order: the wait before the stores is vmcnt(4))
    if (it + PFA < NK) load_a(set, it + PFA);
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      f16x8_t A_[2][2], B_[4][2];
#pragma unroll
      for (int p = 0; p < 2; p++) {
        const unsigned so = (unsigned)(((p * 4 + 2 * ks + kh) ^ sw) << 4);
#pragma unroll
        for (int i = 0; i < 2; i++) A_[i][p] = *reinterpret_cast<const f16x8_t*>(lds + fa + i * 4096 + so);
#pragma unroll
        for (int j = 0; j < 4; j++) B_[j][p] = *reinterpret_cast<const f16x8_t*>(lds + fb + j * 4096 + so);
      }
#pragma unroll
      for (int pp = 0; pp < 3; pp++) {            // small terms first: lo*hi, hi*lo, hi*hi
        const int pa = pp == 0 ? 1 : 0, pb = pp == 1 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[i][pa], B_[j][pb], acc[i][j], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // M: the tile's four 64-column slices are four contiguous 32 KB chunks.  A 32x32 accumulator holds row R in lanes 0..31 and row
  // R + 4 in lanes 32..63; v_permlane32_swap of two neighbouring column tiles puts all 64 columns of one row into one register:
  // every store is one 256-byte run.
  float* mbase = a.Mb + ((((size_t)m_tile * h.npos + pos) * (size_t)(a.Ntot >> 6) + (size_t)n_tile * 4 + wn * 2) * 128 + (size_t)(wm * 64)) * 64 + lane;
#pragma unroll
  for (int qq = 0; qq < 2; qq++)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const unsigned x0 = __float_as_uint(acc[i][2 * qq][r]), x1 = __float_as_uint(acc[i][2 * qq + 1][r]);
        const auto s32 = __builtin_amdgcn_permlane32_swap(x0, x1, false, false);   // [x0 low half, x1 low half], [x0 high half, x1 high half]
        const unsigned w0 = s32[0], w1 = s32[1];
        float* d = mbase + (size_t)qq * 8192 + (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * 64;
        d[0] = __uint_as_float(w0);
        d[4 * 64] = __uint_as_float(w1);
      }
}

// ---- the same GEMM with the operands DMA'd straight into LDS (buffer_load ... lds, 16 bytes per lane) ----------------------------
// Round-4 decomposition of wino_gemm_h2c_kernel (profiles/r04/gemm_decomposition.log): 0.335 ms; A from cache 0.311; no M stores
// 0.256; everything from cache and no stores 0.215 — the kernel is neither HBM- nor MFMA-bound (the three fp16 MFMAs per product
// are 0.126 ms of matrix pipe): a workgroup's prologue, its eight barrier-separated K steps and its 128-stores-per-thread epilogue
// ADD, and two workgroups per CU (224 registers: 128 accumulators + 64 staging + fragments) cannot cover one another.
// Without staging registers (no VGPR destination, no ds_write pass — the LDS image is lane-linear, the swizzle goes on the SOURCE
// address) and with the B fragments read just in time the kernel fits 168 registers: THREE workgroups per CU, single LDS stage
// (48 KB each), plain __syncthreads() around every stage (no DMA is ever in flight across a barrier).
typedef __attribute__((address_space(3))) void* h2c_lds_ptr_t;
// (Measured and dropped, profiles/r04/gemm_l2_prefetch_ab.log: pulling the A chunk of step k + 1..3 into L2 while step k computes —
// one 4-byte buffer_load ... lds per thread and 64 bytes — 0.298-0.315 ms against 0.300-0.304: the HBM round trip of A is not what a
// K step waits for.  What the kernel runs into is the L2 -> CU operand stream: 2.4 GB per launch, 11 TB/s at the all-from-cache
// floor of 0.215 ms — the rate round 3's l2_probe measured for a GEMM's mixed hit/miss stream.)
// NT: M is stored NON-TEMPORALLY (round 5).  M (0.82 GB per launch) is written once and read once, by the next kernel: stored with the
// default policy it displaces the operands this kernel and its neighbour on the other queue re-read from L2 (V2c by both column tiles, the
// weight image by every row tile).  The GEMM itself does not get faster (0.311 vs 0.311 ms), the out->in kernel that follows does (0.286 ->
// 0.270 ms), and with V2c stored the same way by that kernel the two-queue pass goes 12.27 -> 11.95 ms on one box (profiles/r05).
template <int NK, bool NT = true>
__global__ __launch_bounds__(256, 3) void wino_gemm_h2g_kernel(WinoH2Args h) {
  const WinoArgs& a = h.w;
  constexpr int SA = 128 * 128, SB = 256 * 128;
  __shared__ __attribute__((aligned(16))) unsigned char lds[SA + SB];

  const int n_nt = a.Ntot >> 8;
  const int per_pos = a.n_mtiles * n_nt;
  const int nblk = h.npos * per_pos;
  const int id = blockIdx.x;
  int q = nblk >> 3, rr = nblk & 7, xcd = id & 7, slot = id >> 3;
  int tile = (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + slot;
  const int pos = tile / per_pos;
  const int rem = tile - pos * per_pos;
  const int m_tile = rem / n_nt, n_tile = rem - m_tile * n_nt;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid >> 1, wn = wid & 1;
  // DMA: wave w fills rows 32 w .. 32 w + 31 of the A image (4 instructions of 8 rows) and rows 64 w .. + 63 of the B image (8);
  // LDS unit (row r, slot q') receives global unit (r, q' ^ ((r >> 1) & 7)); (r >> 1) & 7 = (4 j + lane / 16) & 7 for instruction j
  const __amdgpu_buffer_rsrc_t ar = h2_rsrc(a.V), br = h2_rsrc(h.U2c);
  const unsigned vo_e = maps::h2c_dma_src(lane, 0), vo_o = maps::h2c_dma_src(lane, 1);   // instructions j even / odd: + (j / 2) * 2048 (scalar)
  const unsigned a_w = maps::h2c_wave_part(wid, 128), b_w = maps::h2c_wave_part(wid, 256);   // the wave's part of a chunk (bytes)
  unsigned a_so = (unsigned)(((size_t)m_tile * h.npos + pos) * (size_t)NK * SA) + a_w;
  unsigned b_so = (unsigned)((((size_t)pos * NK) * n_nt + n_tile) * (size_t)SB) + b_w;
  const unsigned b_step = (unsigned)n_nt * SB;
  unsigned char* const la = lds + a_w;
  unsigned char* const lb = lds + SA + b_w;

  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

#pragma nounroll
  for (int it = 0; it < NK; it++) {
#pragma unroll
    for (int j = 0; j < maps::h2c_wave_instrs(128); j++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(ar, (h2c_lds_ptr_t)(la + maps::h2c_dma_dst(0, j)), 16, (j & 1) ? vo_o : vo_e,
                                               a_so + (maps::h2c_dma_src(0, j) - maps::h2c_dma_src(0, j & 1)), 0, 0);
#pragma unroll
    for (int j = 0; j < maps::h2c_wave_instrs(256); j++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(br, (h2c_lds_ptr_t)(lb + maps::h2c_dma_dst(0, j)), 16, (j & 1) ? vo_o : vo_e,
                                               b_so + (maps::h2c_dma_src(0, j) - maps::h2c_dma_src(0, j & 1)), 0, 0);
    a_so += SA; b_so += b_step;
    __syncthreads();                                  // (its fence waits vmcnt(0): this wave's DMA has landed; then every wave's)
#pragma unroll
    for (int ks = 0; ks < 2; ks++) {
      f16x8_t A_[2][2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        A_[i][0] = *reinterpret_cast<const f16x8_t*>(lds + maps::h2c_frag(wm * 64 + i * 32, lane, 0, ks));
        A_[i][1] = *reinterpret_cast<const f16x8_t*>(lds + maps::h2c_frag(wm * 64 + i * 32, lane, 1, ks));
      }
#pragma unroll
      for (int j = 0; j < 4; j++) {                   // B fragments just in time: 8 registers live instead of 32
        const f16x8_t b0 = *reinterpret_cast<const f16x8_t*>(lds + SA + maps::h2c_frag(wn * 128 + j * 32, lane, 0, ks));
        const f16x8_t b1 = *reinterpret_cast<const f16x8_t*>(lds + SA + maps::h2c_frag(wn * 128 + j * 32, lane, 1, ks));
#pragma unroll
        for (int i = 0; i < 2; i++) {                 // small terms first: lo*hi, hi*lo, hi*hi
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[i][1], b0, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[i][0], b1, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[i][0], b0, acc[i][j], 0, 0, 0);
        }
      }
    }
    __syncthreads();                                  // every wave has read the stage before the next DMA overwrites it
  }

  // M: after v_permlane32_swap of two neighbouring column tiles a register holds all 64 columns of one row (lanes = columns): w0 = row
  // mfma_row(r), w1 = that row + 4; every store instruction writes one 256-byte run of Mc
  float* mbase = a.Mb + maps::mc_index(m_tile, h.npos, pos, a.Ntot >> 6, n_tile * 4 + wn * 2, wm * 64, lane);
#pragma unroll
  for (int qq = 0; qq < 2; qq++)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int r = 0; r < 16; r++) {
        const unsigned x0 = __float_as_uint(acc[i][2 * qq][r]), x1 = __float_as_uint(acc[i][2 * qq + 1][r]);
        const auto s32 = __builtin_amdgcn_permlane32_swap(x0, x1, false, false);
        const unsigned w0 = s32[0], w1 = s32[1];
        float* d = mbase + maps::mc_index(0, 0, 0, 0, qq, i * 32 + maps::mfma_row(r), 0);
        if (NT) {
          __builtin_nontemporal_store(__uint_as_float(w0), d);
          __builtin_nontemporal_store(__uint_as_float(w1), d + maps::mc_index(0, 0, 0, 0, 0, 4, 0));
        } else {
          d[0] = __uint_as_float(w0);
          d[maps::mc_index(0, 0, 0, 0, 0, 4, 0)] = __uint_as_float(w1);
        }
      }
}

// (Measured and dropped, profiles/r04/gemm_256x256_pipelined_ab.log: a 256 x 256 tile — 1.6 instead of 2.4 GB of operands from L2 — on 512
// threads, one workgroup per CU, two 64 KB LDS stages with the DMA of step k + 1 in flight under step k's arithmetic (raw s_barrier +
// counted vmcnt; the compiler's LDS-DMA tracking keeps the stages apart when they are separate __shared__ objects): correct, 0.359 ms
// against 0.296 — one workgroup per CU cannot cover its own prologue and its 128-stores-per-thread epilogue, three can.)

// ---- persistent form, B stationary in REGISTERS (round 5) ------------------------------------------------------------------------
// wino_gemm_h2g_kernel streams both operands per 128 x 256 tile: 2.4 GB per launch through the L2 -> CU path, 1.6 GB of it the
// 25.7 MB weight image re-read 62 times, and its 128-stores-per-thread epilogue is serial with the K loop.  Here a workgroup keeps
// ONE weight slab for many tiles: 128 GEMM columns x K = 256 x (hi | lo) fp16 = 128 KB — not in LDS (that would leave 32 KB for the A
// ring: ~16 KB in flight per CU, a latency-bound stream) but in the MFMA B-fragment REGISTERS of its four waves (64 columns x 256 k x
// 2 pieces = 256 registers per lane; a one-workgroup-per-CU kernel owns all 512).  LDS is then one 16-stage ring of 8 KB A stages
// (64 rows x one 32-channel K step) filled by assembly LDS-DMA D = 10 stages ahead (~72 KB in flight per CU) — the counted
// s_waitcnt vmcnt(N) below rely on gfx9's in-order retirement of a wave's vector-memory operations, stores included.
//   team   = the four workgroups (column slabs 0..3) of one XCD slot group; they walk the SAME (position, 64-row half tile) units
//            in the same order, so three of the four A reads are L2 hits: L2 -> CU 1.64 GB (A) + 0.1 GB (B), HBM as before
//   tile   = 64 rows x 128 columns per workgroup, 32 x 64 per wave (two 32x32 accumulators); two accumulator sets: the M stores of
//            tile t (the same permlane32_swap 256-byte runs) are issued four per K step between the MFMAs of tile t + 1
//   units  = npos x 2 n_mtiles half tiles, split evenly over the teams (19x19, B = 512: 6272 units / 64 teams = 98 each, no tail)
// Per accumulator the MFMA sequence (K steps ascending; lo*hi, hi*lo, hi*hi) is wino_gemm_h2g_kernel's: M is bit-identical.
typedef unsigned h2p_rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void h2p_dma16(h2p_rsrc_t rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
  unsigned keep;                                   // (M0 is put back: the compiler does not model assembly writes to it)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
// the two DMA instructions of a wave's part of a stage: one save / restore of M0
__device__ __forceinline__ void h2p_dma16x2(h2p_rsrc_t rsrc, unsigned lds_a, unsigned lds_b, unsigned voff_a, unsigned voff_b, unsigned soff) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, %6 offen lds\n\t"
               "s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "s"(lds_a), "s"(lds_b), "v"(voff_a), "v"(voff_b), "s"(rsrc), "s"(soff) : "memory");
}
#define H2P_WAIT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
// MODE (decomposition runs only, agz_net_set_wino_h2_gemm(net, 2 + 16 * MODE)): bit 0 = no M stores, bit 1 = no DMA (the ring holds whatever LDS held)
template <int MODE>
__global__ __launch_bounds__(256, 1) void wino_gemm_h2p_kernel(WinoH2Args h) {
  const WinoArgs& a = h.w;
  constexpr int NK = maps::H2P_NK, R = maps::H2P_R, SA = 64 * 128, D = maps::H2P_D;   // (ring / look-ahead arithmetic: gemm_maps.hpp)
  __shared__ __attribute__((aligned(1024))) unsigned char lds[R * SA];   // 128 KB

  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wid >> 1, wn = wid & 1;
  const int n_slabs = a.Ntot >> 7, n_nt = a.Ntot >> 8;
  const maps::H2pTeam tm = maps::h2p_team((int)blockIdx.x, (int)gridDim.x, n_slabs);
  if (tm.idle) return;
  const int team = tm.team, slab = tm.slab, nteams = tm.nteams;
  // units are dealt in PAIRS (n_half is even): a team's range starts at an even unit and holds an even number, so the two tiles
  // of a pair (ring halves / accumulator sets 0 and 1) always share a position
  const int n_half = a.n_mtiles * 2, U2 = h.npos * a.n_mtiles;
  const int u0 = maps::h2p_u0(team, nteams, U2), nT = maps::h2p_u0(team + 1, nteams, U2) - u0;
  if (nT <= 0) return;

  h2p_rsrc_t ar;
  {
    const unsigned long long u = (unsigned long long)a.V;
    ar[0] = __builtin_amdgcn_readfirstlane((unsigned)u); ar[1] = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32) & 0xffffu);
    ar[2] = 0x7fffffffu; ar[3] = 0x00020000u;
  }
  // DMA of an 8 KB stage: wave w fills rows 16 w .. 16 w + 15 (2 instructions of 8 rows); LDS unit (row r, slot q') receives global
  // unit (r, q' ^ ((r >> 1) & 7)) — h2c_img's swizzle on the SOURCE address, (r >> 1) & 7 = (4 j + lane / 16) & 7 for instruction j
  const unsigned vo_e = maps::h2c_dma_src(lane, 0), vo_o = maps::h2c_dma_src(lane, 1);
  static_assert(maps::h2c_wave_instrs(64) == 2, "two DMA instructions per wave and stage");
  const unsigned a_w = maps::h2c_wave_part(wid, 64);
  const unsigned dma_l = (unsigned)(size_t)(h2c_lds_ptr_t)lds + a_w;
  auto issue = [&](unsigned vb, int kk, int rslot) {
    if (MODE & 2) return;
    const unsigned so = vb + (unsigned)kk * 16384u + a_w;
    h2p_dma16x2(ar, dma_l + (unsigned)rslot * SA + maps::h2c_dma_dst(0, 0), dma_l + (unsigned)rslot * SA + maps::h2c_dma_dst(0, 1), vo_e, vo_o, so);
  };
  const int kh = lane >> 5;
  auto frag = [&](int rslot, int ks, int p) -> f16x8_t {
    return *reinterpret_cast<const f16x8_t*>(lds + rslot * SA + maps::h2c_frag(wm * 32, lane, p, ks));
  };

  // B fragments of this wave's 64 columns, every K step: Bf[kk][ks][j][piece]
  f16x8_t Bf[NK][2][2][2];
  auto load_b = [&](int pos) {
    const char* bb = reinterpret_cast<const char*>(h.U2c) + ((size_t)pos * NK * n_nt + (slab >> 1)) * 32768 +
                     (size_t)(((slab & 1) * 128 + wn * 64 + (lane & 31)) * 128 + kh * 16);
#pragma unroll
    for (int kk = 0; kk < NK; kk++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int p = 0; p < 2; p++)
            Bf[kk][ks][j][p] = *reinterpret_cast<const f16x8_t*>(bb + (size_t)kk * n_nt * 32768 + j * 4096 + (p * 4 + 2 * ks) * 16);
    // the slab lives in the 256 accumulation registers (MFMA reads B operands from there), everything else in the 256 vector
    // registers: left to itself the allocator mixes the two files and shuffles ~20 registers per K step between them
#pragma unroll
    for (int kk = 0; kk < NK; kk++)
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int j = 0; j < 2; j++)
#pragma unroll
          for (int p = 0; p < 2; p++) asm volatile("" : "+a"(Bf[kk][ks][j][p]));
  };
  // (M stores WITHOUT the lane swap — two whole 128-byte lines per instruction straight from the accumulation registers, 288 fewer
  //  instructions per tile pair — measured SLOWER: 0.325 vs 0.312 ms; the 256-byte runs matter to the memory system, profiles/r05)
  auto m_base = [&](int pos, int hm) -> float* {
    return a.Mb + maps::mc_index(hm >> 1, h.npos, pos, a.Ntot >> 6, slab * 2 + wn, (hm & 1) * 64 + wm * 32, lane);
  };
  auto store_rows = [&](const f32x16 (&ac)[2], float* mb, int r) {     // accumulator register r of both column tiles: rows R and R + 4
    const unsigned x0 = __float_as_uint(ac[0][r]), x1 = __float_as_uint(ac[1][r]);
    const auto s32 = __builtin_amdgcn_permlane32_swap(x0, x1, false, false);
    const unsigned w0 = s32[0], w1 = s32[1];
    float* d = mb + maps::mc_index(0, 0, 0, 0, 0, maps::mfma_row(r), 0);
    if (MODE & 1) { asm volatile("" ::"v"(w0), "v"(w1)); return; }
    __builtin_nontemporal_store(__uint_as_float(w0), d);          // (non-temporal, as wino_gemm_h2g_kernel's)
    __builtin_nontemporal_store(__uint_as_float(w1), d + maps::mc_index(0, 0, 0, 0, 0, 4, 0));
  };

  int pos = u0 / n_half;
  // byte offset in V2c of tile tt's K step 0, tt within two tiles of the current position's run (tt clamped: past the end the ring
  // re-fetches the last unit — the counted waits need the same number of operations in every step)
  auto unit_base = [&](int tt) -> unsigned {
    int m = u0 + (tt < nT ? tt : nT - 1) - pos * n_half, p = pos;
    if (m >= n_half) { m -= n_half; p++; }
    return maps::h2p_v_base(p, m, h.npos);
  };
  {                                                          // prologue: stages 0 .. D - 1 (units 0 and 1)
    const unsigned vb0 = unit_base(0), vb1 = unit_base(1);
#pragma unroll
    for (int g = 0; g < D; g++) issue(g < NK ? vb0 : vb1, g % NK, maps::h2p_slot(g / NK, g % NK));
  }
  f32x16 accA[2], accB[2];
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int r = 0; r < 16; r++) accB[j][r] = 0.f;
  f16x8_t Fc[2][2];                                          // A fragments of the step about to run: [ks][piece]
  H2P_WAIT(maps::h2p_wait_early() + 2);                      // stage 0 has landed (this wave's part; after the barrier every wave's)
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int ks = 0; ks < 2; ks++)
#pragma unroll
    for (int p = 0; p < 2; p++) Fc[ks][p] = frag(maps::h2p_slot(0, 0), ks, p);
  // tile 0 has no predecessor: its store slots write the zeros of accB to its OWN rows (overwritten by its results during tile 1:
  // a wave's stores to one address stay in order) — every step of the kernel then issues the same 2 DMAs + 4 stores
  float* mprev = m_base(pos, u0 - pos * n_half);

  // one tile: PAR = t & 1 selects the ring half of its own stages (compile-time LDS offsets) and the accumulator set
  auto tile = [&](auto par_c, int t, f32x16 (&accC)[2], f32x16 (&accP)[2]) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    const int hm = u0 + t - pos * n_half;
    const unsigned vb1 = unit_base(t + 1), vb2 = unit_base(t + 2);
    float* const mcur = m_base(pos, hm);
    const unsigned early = (unsigned)__builtin_amdgcn_readfirstlane((maps::h2p_early(t) || (MODE & 1)) ? 1 : 0);
    f32x16 zero16;                                             // the tile's first product takes a zero C operand (an inline constant: nothing to clear)
#pragma unroll
    for (int r = 0; r < 16; r++) zero16[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < NK; kk++) {
      // stage g + 1 has landed: younger operations than its two DMAs = 4 stores of the issuing step + 8 steps x (2 DMA + 4 stores)
      // = 52; in the first pair the younger operations are prologue DMAs: >= (D - 2) x 2 (a count that is too small only waits longer)
      asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 .Lh2p_steady_%=\n\ts_waitcnt vmcnt(%1)\n\ts_branch .Lh2p_done_%=\n"
                   ".Lh2p_steady_%=:\n\ts_waitcnt vmcnt(%2)\n.Lh2p_done_%=:" ::"s"(early), "n"(maps::h2p_wait_early()), "n"(maps::h2p_wait_steady()) : "memory", "scc");
      __builtin_amdgcn_s_barrier();
      issue(maps::h2p_tiles_ahead(kk) == 1 ? vb1 : vb2, maps::h2p_kk_ahead(kk), maps::h2p_slot_ahead(PAR, kk));
      f16x8_t Fn[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int p = 0; p < 2; p++) Fn[ks][p] = frag(maps::h2p_slot_next(PAR, kk), ks, p);
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {                       // small terms first: lo*hi, hi*lo, hi*hi (per accumulator as h2g)
        accC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Fc[ks][1], Bf[kk][ks][0][0], (kk | ks) ? accC[0] : zero16, 0, 0, 0);
        accC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Fc[ks][1], Bf[kk][ks][1][0], (kk | ks) ? accC[1] : zero16, 0, 0, 0);
        accC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Fc[ks][0], Bf[kk][ks][0][1], accC[0], 0, 0, 0);
        accC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Fc[ks][0], Bf[kk][ks][1][1], accC[1], 0, 0, 0);
        store_rows(accP, mprev, 2 * kk + ks);                // the previous tile's M, two row groups per K step
        accC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Fc[ks][0], Bf[kk][ks][0][0], accC[0], 0, 0, 0);
        accC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Fc[ks][0], Bf[kk][ks][1][0], accC[1], 0, 0, 0);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int p = 0; p < 2; p++) Fc[ks][p] = Fn[ks][p];
    }
    mprev = mcur;
  };
  for (int t = 0; t < nT;) {                                 // a run of tiles at one position: its weight slab into the B registers
    const int seg_end = min(nT, (pos + 1) * n_half - u0);
    load_b(pos);
    for (; t < seg_end; t += 2) {
      tile(std::integral_constant<int, 0>{}, t, accA, accB);
      tile(std::integral_constant<int, 1>{}, t + 1, accB, accA);
    }
    pos++;
  }
#pragma unroll
  for (int r = 0; r < 16; r++) store_rows(accB, mprev, r);   // the last tile (odd)
  H2P_WAIT(0);                                               // no DMA may still be landing when the LDS goes back to the CU
}

// Y[l] += At[l][nu] * t for the output transform's row pass, one column nu at a time (nu is a constant after unrolling)
template <int TM> __device__ __forceinline__ void wino_at_acc(int nu, float* Y, float t);
template <> __device__ __forceinline__ void wino_at_acc<5>(int nu, float* Y, float t) {
  switch (nu) {
    case 0: Y[0] += t; break;
    case 1: Y[0] += t; Y[1] += t; Y[2] += t; Y[3] += t; Y[4] += t; break;
    case 2: Y[0] += t; Y[1] -= t; Y[2] += t; Y[3] -= t; Y[4] += t; break;
    case 3: Y[0] += t; Y[1] += 0.5f * t; Y[2] += 0.25f * t; Y[3] += 0.125f * t; Y[4] += 0.0625f * t; break;
    case 4: Y[0] += t; Y[1] -= 0.5f * t; Y[2] += 0.25f * t; Y[3] -= 0.125f * t; Y[4] += 0.0625f * t; break;
    case 5: Y[0] += t; Y[1] += 2.f * t; Y[2] += 4.f * t; Y[3] += 8.f * t; Y[4] += 16.f * t; break;
    default: Y[4] += t; break;
  }
}
template <> __device__ __forceinline__ void wino_at_acc<4>(int nu, float* Y, float t) {
  switch (nu) {
    case 0: Y[0] += t; break;
    case 1: Y[0] += t; Y[1] += t; Y[2] += t; Y[3] += t; break;
    case 2: Y[0] += t; Y[1] -= t; Y[2] += t; Y[3] -= t; break;
    case 3: Y[0] += t; Y[1] += 2.f * t; Y[2] += 4.f * t; Y[3] += 8.f * t; break;
    case 4: Y[0] += t; Y[1] -= 2.f * t; Y[2] += 4.f * t; Y[3] -= 8.f * t; break;
    default: Y[3] += t; break;
  }
}

// Output transform of block l + block epilogue + (LAST ? y to HBM : input transform of block l+1).
// grid (T / 16 rounded up, C / 32), 256 threads: thread = (tile row tl = tid / 16 of the group, channel pair pr = tid % 16 of the slice);
// a 16-lane group is one tile, a wave four consecutive tile rows.  Requires 16 % TPB == 0 (a group holds whole boards).
// Dynamic LDS (not LAST): [16 / TPB boards][H * W][32] fp32.
template <int TM, bool LAST>
__global__ __launch_bounds__(256, 2) void wino_oi_h2c_kernel(WinoH2Args h) {
  using WT = WinoT<TM>;
  constexpr int AL = WT::AL;
  const WinoArgs& a = h.w;
  extern __shared__ __attribute__((aligned(16))) float ys[];
  const int tid = threadIdx.x, tl = tid >> 4, pr = tid & 15, lane = tid & 63;
  const int s = blockIdx.y, NS = a.C >> 5, HW = a.H * a.W;
  const int t0 = blockIdx.x * 16;                      // uniform; t0 % 16 == 0, so t0 >> 7 is the 128-row tile of the whole group
  const int t = t0 + tl;
  const bool live = t < a.T;
  const int tc = live ? t : a.T - 1;
  const int b = tc / a.TPB, tt = tc - b * a.TPB;
  const int ty = tt / a.ntx, tx = tt - ty * a.ntx;
  const int bl = tl / a.TPB;                           // board of the group
  float s_, un0;
  wino_h2_scales(h.amax_in[b], WT::VSHIFT, &s_, &un0);
  // ---- output transform: stream the AL columns of the position grid (AL float4 {a c0, b c0, a c1, b c1} each), column pass, accumulate the row pass
  const __amdgpu_buffer_rsrc_t mr = h2_rsrc(a.Mb + ((size_t)(t0 >> 7) * h.npos * NS + s) * 8192);
  const unsigned m_lane = (unsigned)(((tc & 127) * 64 + pr * 4) * 4);
  const unsigned m_pos = (unsigned)NS * 32768u;        // bytes between positions
  float Y[TM][TM][4];
#pragma unroll
  for (int k = 0; k < TM; k++)
#pragma unroll
    for (int l = 0; l < TM; l++)
#pragma unroll
      for (int e = 0; e < 4; e++) Y[k][l][e] = 0.f;
  float4 m[2][AL];
#pragma unroll
  for (int xi = 0; xi < AL; xi++) m[0][xi] = h2_ldf4(mr, m_lane, (unsigned)(xi * AL) * m_pos);
#pragma unroll
  for (int nu = 0; nu < AL; nu++) {
    if (nu + 1 < AL) {
#pragma unroll
      for (int xi = 0; xi < AL; xi++) m[(nu + 1) & 1][xi] = h2_ldf4(mr, m_lane, (unsigned)(xi * AL + nu + 1) * m_pos);
    }
    float mm[4][AL], oo[4][TM];
#pragma unroll
    for (int xi = 0; xi < AL; xi++) { const float4 v = m[nu & 1][xi]; mm[0][xi] = v.x; mm[1][xi] = v.y; mm[2][xi] = v.z; mm[3][xi] = v.w; }
#pragma unroll
    for (int e = 0; e < 4; e++) wino_atv<TM>(mm[e], oo[e]);
#pragma unroll
    for (int k = 0; k < TM; k++)
#pragma unroll
      for (int e = 0; e < 4; e++) {
        float yl[TM];
#pragma unroll
        for (int l = 0; l < TM; l++) yl[l] = Y[k][l][e];
        wino_at_acc<TM>(nu, yl, oo[e][k]);
#pragma unroll
        for (int l = 0; l < TM; l++) Y[k][l][e] = yl[l];
      }
    __builtin_amdgcn_sched_barrier(0);   // keep the prefetch one column deep (the compiler would hoist all AL^2 loads and spill)
  }
  // ---- block epilogue (parameters float4 {sa, ta, sb, tb} per (pixel, channel), scales folded in: agz_net::build_wino_h2_weights)
  const char* ep = reinterpret_cast<const char*>(a.ep);
  const unsigned e_lane = (unsigned)(32 * s + 2 * pr) * 16u, e_pix = (unsigned)a.Cout_p * 16u;
  float mx = 0.f;
  float* yb = LAST ? a.y + (size_t)b * a.Hp * a.Wp * a.Cout_p + 32 * s + 2 * pr : nullptr;
#pragma unroll
  for (int k = 0; k < TM; k++) {
    const int hh = TM * ty + k, hc = hh < a.H ? hh : a.H - 1;
#pragma unroll
    for (int l = 0; l < TM; l++) {
      const int ww = TM * tx + l, wc = ww < a.W ? ww : a.W - 1;
      const char* e = ep + (size_t)(hc * a.W + wc) * e_pix + e_lane;
      const float4 E0 = *reinterpret_cast<const float4*>(e), E1 = *reinterpret_cast<const float4*>(e + 16);
      float va = (Y[k][l][0] * un0) * E0.x + E0.y, vb = (Y[k][l][1] * un0) * E0.z + E0.w;
      float vc = (Y[k][l][2] * un0) * E1.x + E1.y, vd = (Y[k][l][3] * un0) * E1.z + E1.w;
      va = va > 0.f ? va : 0.f; vb = vb > 0.f ? vb : 0.f; vc = vc > 0.f ? vc : 0.f; vd = vd > 0.f ? vd : 0.f;
      const float y0 = va + vb, y1 = vc + vd;          // relu(a) + relu(b) >= 0 already
      if (live && hh < a.H && ww < a.W) {
        if (LAST) *reinterpret_cast<float2*>(yb + ((size_t)(hh + 1) * a.Wp + (ww + 1)) * a.Cout_p) = make_float2(y0, y1);
        else *reinterpret_cast<float2*>(&ys[((size_t)bl * HW + hh * a.W + ww) * 32 + 2 * pr]) = make_float2(y0, y1);
        mx = fmaxf(mx, fmaxf(y0, y1));
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // one row of parameter loads in flight (all TM^2 hoisted: spills)
  }
  if (LAST) return;
  // the tile's maximum over this slice -> the next consumer's exact max|x|
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if (pr == 0 && live) h.wm_out[(size_t)t * NS + s] = mx;
  // ---- range of V2(l+1): the commit-time bound on max |y| from the exact maximum of this block's input
  float ax;
  if (h.amax_true) ax = __uint_as_float(reinterpret_cast<const unsigned*>(h.amax_true)[b]);
  else {
    const int wmpb = a.TPB * NS;
    const float* wp = h.wm_prev + (size_t)b * wmpb;
    ax = 0.f;
    for (int i = pr; i < wmpb; i += 16) ax = fmaxf(ax, wp[i]);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) ax = fmaxf(ax, __shfl_xor(ax, o, 64));
  }
  const float bound = h.g1 * ax + h.g0;
  const unsigned bbits = __float_as_uint(bound);
  if (s == 0 && tt == 0 && pr == 0 && live) h.amax_next[b] = bbits;
  float sb, inv_;
  wino_h2_scales(bbits, WT::VSHIFT, &sb, &inv_);
  __syncthreads();
  // ---- input transform of block l+1 on the LDS board: tile t, channels 32 s + 2 pr, + 1 (conv_wino_h2.hpp's wino_in_h2_kernel)
  const float* yl = ys + (size_t)bl * HW * 32 + 2 * pr;
  float tmx[AL][AL], tmy[AL][AL];
#pragma unroll
  for (int j = 0; j < AL; j++) {
    const int ww = TM * tx + j - 1;
    const bool okx = ww >= 0 && ww < a.W;
    const int wc = ww < 0 ? 0 : (ww < a.W ? ww : a.W - 1);
    float dx[AL], dy[AL], ox[AL], oy[AL];
#pragma unroll
    for (int i = 0; i < AL; i++) {
      const int hh = TM * ty + i - 1;
      const bool ok = okx && hh >= 0 && hh < a.H;
      const int hc = hh < 0 ? 0 : (hh < a.H ? hh : a.H - 1);
      const float2 d = *reinterpret_cast<const float2*>(yl + (size_t)(hc * a.W + wc) * 32);
      dx[i] = ok ? d.x : 0.f; dy[i] = ok ? d.y : 0.f;
    }
    wino_btv<TM>(dx, ox);
    wino_btv<TM>(dy, oy);
#pragma unroll
    for (int i = 0; i < AL; i++) { tmx[i][j] = ox[i]; tmy[i][j] = oy[i]; }
  }
  const float sv = live ? sb : 0.f;                    // rows past the last tile: zeros
  // V2c: the group's 16 rows of K step s are 2 KB; a wave's four rows 512 B = two 256-byte stores (hi | lo of two rows each)
  const __amdgpu_buffer_rsrc_t vr = h2_rsrc(reinterpret_cast<const char*>(a.V) + ((size_t)(t0 >> 7) * h.npos * NS + s) * 16384);
  const unsigned v_lane = (unsigned)(((t0 & 127) + (tid >> 6) * 4) * 128 + lane * 4);
  const unsigned v_pos = (unsigned)NS * 16384u;
#pragma unroll
  for (int i = 0; i < AL; i++) {
    float ox[AL], oy[AL];
    wino_btv<TM>(tmx[i], ox);
    wino_btv<TM>(tmy[i], oy);
#pragma unroll
    for (int j = 0; j < AL; j++) {
      unsigned lo;
      const unsigned hi = wino_h2_pack(ox[j] * sv, oy[j] * sv, &lo);
      const auto s16 = __builtin_amdgcn_permlane16_swap(hi, lo, false, false);   // [hi t0, lo t0, hi t2, lo t2], [hi t1, lo t1, hi t3, lo t3]
      const unsigned e0 = s16[0], e1 = s16[1];
      const auto s32 = __builtin_amdgcn_permlane32_swap(e0, e1, false, false);   // rows t0, t1 | rows t2, t3
      const unsigned w0 = s32[0], w1 = s32[1];
      __builtin_amdgcn_raw_buffer_store_b32(w0, vr, v_lane, (unsigned)(i * AL + j) * v_pos, 0);
      __builtin_amdgcn_raw_buffer_store_b32(w1, vr, v_lane + 256u, (unsigned)(i * AL + j) * v_pos, 0);
    }
  }
}

// ---- the pipelined form of the same kernel --------------------------------------------------------------------------------------
// What the measurements of round 4 said about wino_oi_h2c_kernel (profiles/r04/oi_decomposition.log): 0.364 ms per headline block against
// 0.359 for the two kernels it replaces, although it moves 0.38 GB less.  With the M loads served from L1 0.176, without the parameter
// loads 0.31, without the V2 stores 0.335, with none of the three 0.111: the phases of a workgroup ADD (load M, 23 us; parameters,
// 7 us; transform + store, 14 us), two workgroups per CU are all the registers allow (100 accumulators + 14 loads in flight), and
// putting the CU's two workgroups half a period apart changes nothing — each workgroup's own chain of round trips is the time.
// Here a workgroup is persistent and the next item's M columns are in flight while this item's input transform runs:
//  * the input transform recomputes its column pass per output row from a ZERO-HALOED LDS board (every read a ds_read_b64 with an
//    immediate offset, no bounds arithmetic): ~70 live registers instead of ~200, which leaves room for NPRE = AL - 2 columns
//    (35 float4 at F(5x5,3x3)) of the next item's M to land during it; the last two columns are fetched under the first ones' arithmetic;
//  * the epilogue's parameter rows are fetched one row ahead;
//  * transform arithmetic on the packed-fp32 pipe (two channels / two branches per instruction).
// Item = (16-tile-row group g, slice s), taken in the order item = g * NS + s, workgroup w takes items w, w + grid, ...
typedef float f2c __attribute__((ext_vector_type(2)));
template <int TM> __device__ __forceinline__ void wino_atv_p(const f2c* m, f2c* o);
template <> __device__ __forceinline__ void wino_atv_p<4>(const f2c* m, f2c* o) {
  const f2c s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  o[0] = m[0] + s12 + s34;
  o[1] = d12 + 2.f * d34;
  o[2] = s12 + 4.f * s34;
  o[3] = d12 + 8.f * d34 + m[5];
}
template <> __device__ __forceinline__ void wino_atv_p<5>(const f2c* m, f2c* o) {
  const f2c s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
  o[0] = m[0] + s12 + s34 + m[5];
  o[1] = d12 + 0.5f * d34 + 2.f * m[5];
  o[2] = s12 + 0.25f * s34 + 4.f * m[5];
  o[3] = d12 + 0.125f * d34 + 8.f * m[5];
  o[4] = s12 + 0.0625f * s34 + 16.f * m[5] + m[6];
}
template <int TM> __device__ __forceinline__ void wino_at_acc_p(int nu, f2c* Y, f2c t);
template <> __device__ __forceinline__ void wino_at_acc_p<5>(int nu, f2c* Y, f2c t) {
  switch (nu) {
    case 0: Y[0] += t; break;
    case 1: Y[0] += t; Y[1] += t; Y[2] += t; Y[3] += t; Y[4] += t; break;
    case 2: Y[0] += t; Y[1] -= t; Y[2] += t; Y[3] -= t; Y[4] += t; break;
    case 3: Y[0] += t; Y[1] += 0.5f * t; Y[2] += 0.25f * t; Y[3] += 0.125f * t; Y[4] += 0.0625f * t; break;
    case 4: Y[0] += t; Y[1] -= 0.5f * t; Y[2] += 0.25f * t; Y[3] -= 0.125f * t; Y[4] += 0.0625f * t; break;
    case 5: Y[0] += t; Y[1] += 2.f * t; Y[2] += 4.f * t; Y[3] += 8.f * t; Y[4] += 16.f * t; break;
    default: Y[4] += t; break;
  }
}
template <> __device__ __forceinline__ void wino_at_acc_p<4>(int nu, f2c* Y, f2c t) {
  switch (nu) {
    case 0: Y[0] += t; break;
    case 1: Y[0] += t; Y[1] += t; Y[2] += t; Y[3] += t; break;
    case 2: Y[0] += t; Y[1] -= t; Y[2] += t; Y[3] -= t; break;
    case 3: Y[0] += t; Y[1] += 2.f * t; Y[2] += 4.f * t; Y[3] += 8.f * t; break;
    case 4: Y[0] += t; Y[1] -= 2.f * t; Y[2] += 4.f * t; Y[3] -= 8.f * t; break;
    default: Y[3] += t; break;
  }
}
template <int TM> __device__ __forceinline__ f2c wino_bt_row_p(int i, const f2c* d);
template <> __device__ __forceinline__ f2c wino_bt_row_p<5>(int i, const f2c* d) {
  switch (i) {
    case 0: return -0.5f * d[0] + 0.25f * d[1] + 2.5f * d[2] - 1.25f * d[3] - 2.f * d[4] + d[5];
    case 1: return 0.5f * d[1] + 0.25f * d[2] - 2.25f * d[3] - d[4] + d[5];
    case 2: return -0.5f * d[1] + 0.75f * d[2] + 1.75f * d[3] - 3.f * d[4] + d[5];
    case 3: return d[1] + 1.5f * d[2] - 2.f * d[3] - 1.5f * d[4] + d[5];
    case 4: return -d[1] + 2.5f * (d[2] - d[4]) + d[5];
    case 5: return 0.25f * d[1] - 1.25f * d[3] + d[5];
    default: return -0.5f * d[1] + 0.25f * d[2] + 2.5f * d[3] - 1.25f * d[4] - 2.f * d[5] + d[6];
  }
}
template <> __device__ __forceinline__ f2c wino_bt_row_p<4>(int i, const f2c* d) {
  switch (i) {
    case 0: return 4.f * d[0] - 5.f * d[2] + d[4];
    case 1: return -4.f * d[1] - 4.f * d[2] + d[3] + d[4];
    case 2: return 4.f * d[1] - 4.f * d[2] - d[3] + d[4];
    case 3: return -2.f * d[1] - d[2] + 2.f * d[3] + d[4];
    case 4: return 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
    default: return 4.f * d[1] - 5.f * d[3] + d[5];
  }
}
template <int TM> __device__ __forceinline__ void wino_btv_p(const f2c* d, f2c* o) {
#pragma unroll
  for (int i = 0; i < TM + 2; i++) o[i] = wino_bt_row_p<TM>(i, d);
}

// grid: any number of workgroups <= items (2 per CU); 256 threads; dynamic LDS [16 / TPB boards][TM nty + 2][TM ntx + 2][32] fp32
// NT: V2c stored non-temporally (see wino_gemm_h2g_kernel; M loads non-temporal as well measured no better: 11.99 vs 11.95 ms per pass)
template <int TM, bool NT = true>
__global__ __launch_bounds__(256, 2) void wino_oip_h2c_kernel(WinoH2Args h) {
  constexpr int MAUX = 0, VAUX = NT ? 2 : 0;
  using WT = WinoT<TM>;
  constexpr int AL = WT::AL, NPRE = AL - 3;
  const WinoArgs& a = h.w;
  extern __shared__ __attribute__((aligned(16))) float ys[];
  const int tid = threadIdx.x, tl = tid >> 4, pr = tid & 15, lane = tid & 63;
  const int NS = a.C >> 5;
  const int PH = TM * a.nty + 2, PW = TM * a.ntx + 2;        // haloed board: pixel (hh, ww) at (hh + 1, ww + 1)
  const int bpg = 16 / a.TPB;                                // boards per group
  const int n_groups = (a.T + 15) >> 4, n_items = n_groups * NS;
  const int tt = tl % a.TPB, bl = tl / a.TPB;
  const int ty = tt / a.ntx, tx = tt - ty * a.ntx;
  // zero the LDS boards once: the halo and the cells past H x W are never written afterwards
  for (int i = tid; i < bpg * PH * PW * 32; i += 256) ys[i] = 0.f;
  const __amdgpu_buffer_rsrc_t mr = h2_rsrc(a.Mb);
  const __amdgpu_buffer_rsrc_t vr = h2_rsrc(a.V);
  const unsigned m_lane = (unsigned)((tl * 64 + pr * 4) * 4);
  const unsigned v_lane = (unsigned)((((tid >> 6) * 4) * 128) + lane * 4);
  const unsigned e_pix = (unsigned)a.Cout_p * 16u;
  // this thread's tile origin in the haloed LDS board (floats): output pixel (k, l) at yo + ((k + 1) * PW + l + 1) * 32, input patch (i, j) at yi + (i * PW + j) * 32
  float* const ybase = ys + ((size_t)bl * PH * PW + (size_t)(TM * ty) * PW + TM * tx) * 32 + 2 * pr;
  unsigned okmask = 0;                                       // bit k * TM + l: output pixel (k, l) of this thread's tile lies on the board
#pragma unroll
  for (int k = 0; k < TM; k++)
#pragma unroll
    for (int l = 0; l < TM; l++) okmask |= (TM * ty + k < a.H && TM * tx + l < a.W) ? 1u << (k * TM + l) : 0u;
  const __amdgpu_buffer_rsrc_t er = h2_rsrc(a.ep);
  float4 m[NPRE][AL];
  auto m_soff = [&](int g, int s, int pos) -> unsigned { return (unsigned)((((g >> 3) * h.npos + pos) * NS + s)) * 32768u + (unsigned)(g & 7) * 4096u; };
  int item = blockIdx.x;
  {
    const int g = __builtin_amdgcn_readfirstlane(item / NS), s = __builtin_amdgcn_readfirstlane(item - (item / NS) * NS);
#pragma unroll
    for (int nu = 0; nu < NPRE; nu++)
#pragma unroll
      for (int xi = 0; xi < AL; xi++) m[nu][xi] = h2_ldf4<MAUX>(mr, m_lane, m_soff(g, s, xi * AL + nu));
  }
  __syncthreads();
  for (; item < n_items; item += gridDim.x) {
    const int g = __builtin_amdgcn_readfirstlane(item / NS), s = __builtin_amdgcn_readfirstlane(item - (item / NS) * NS);
    const int t = g * 16 + tl;
    const bool live = t < a.T;
    const int b = live ? g * bpg + bl : (a.T - 1) / a.TPB;
    float s_, un0;
    wino_h2_scales(h.amax_in[b], WT::VSHIFT, &s_, &un0);
    // ---- output transform: columns 0 .. NPRE-1 are in registers (or landing); column nu + NPRE is fetched into column nu's registers
    f2c Y[TM][TM][2];
#pragma unroll
    for (int k = 0; k < TM; k++)
#pragma unroll
      for (int l = 0; l < TM; l++) { Y[k][l][0] = f2c{0.f, 0.f}; Y[k][l][1] = f2c{0.f, 0.f}; }
#pragma unroll
    for (int nu = 0; nu < AL; nu++) {
      f2c c0[AL], c1[AL], o0[TM], o1[TM];
#pragma unroll
      for (int xi = 0; xi < AL; xi++) {
        const float4 v = m[nu % NPRE][xi];
        c0[xi] = f2c{v.x, v.y}; c1[xi] = f2c{v.z, v.w};
      }
      if (nu + NPRE < AL) {
#pragma unroll
        for (int xi = 0; xi < AL; xi++) m[nu % NPRE][xi] = h2_ldf4<MAUX>(mr, m_lane, m_soff(g, s, xi * AL + nu + NPRE));
      }
      wino_atv_p<TM>(c0, o0);
      wino_atv_p<TM>(c1, o1);
#pragma unroll
      for (int k = 0; k < TM; k++) {
        f2c y0[TM], y1[TM];
#pragma unroll
        for (int l = 0; l < TM; l++) { y0[l] = Y[k][l][0]; y1[l] = Y[k][l][1]; }
        wino_at_acc_p<TM>(nu, y0, o0[k]);
        wino_at_acc_p<TM>(nu, y1, o1[k]);
#pragma unroll
        for (int l = 0; l < TM; l++) { Y[k][l][0] = y0[l]; Y[k][l][1] = y1[l]; }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- block epilogue, parameter rows one row ahead.  Addresses are built from one per-iteration base (made opaque to the
    // compiler: hoisted out of the item loop the 2 TM^2 parameter addresses alone hold 50 registers across every phase); pixels past
    // the board read the tile's first pixel and store a zero (the LDS cells past H x W stay the next transform's zero padding)
    unsigned e_base = (unsigned)(32 * s + 2 * pr) * 16u + (unsigned)((TM * ty) * a.W + TM * tx) * e_pix;
    asm volatile("" : "+v"(e_base));
    float4 E[2][TM][2];
    auto load_e = [&](int buf, int k) {
#pragma unroll
      for (int l = 0; l < TM; l++) {
        const bool ok = (okmask >> (k * TM + l)) & 1u;
        const unsigned off = e_base + (ok ? (unsigned)(k * a.W + l) * e_pix : 0u);
        E[buf][l][0] = h2_ldf4(er, off, 0);
        E[buf][l][1] = h2_ldf4(er, off + 16u, 0);
      }
    };
    load_e(0, 0);
    float mxv = 0.f;
    float* yw = ybase;
    asm volatile("" : "+v"(yw));
#pragma unroll
    for (int k = 0; k < TM; k++) {
      if (k + 1 < TM) load_e((k + 1) & 1, k + 1);
#pragma unroll
      for (int l = 0; l < TM; l++) {
        const bool ok = live && ((okmask >> (k * TM + l)) & 1u);
        const float4 E0 = E[k & 1][l][0], E1 = E[k & 1][l][1];
        const f2c u0 = Y[k][l][0] * un0, u1 = Y[k][l][1] * un0;
        float va = u0.x * E0.x + E0.y, vb = u0.y * E0.z + E0.w;
        float vc = u1.x * E1.x + E1.y, vd = u1.y * E1.z + E1.w;
        va = va > 0.f ? va : 0.f; vb = vb > 0.f ? vb : 0.f; vc = vc > 0.f ? vc : 0.f; vd = vd > 0.f ? vd : 0.f;
        const float y0 = ok ? va + vb : 0.f, y1 = ok ? vc + vd : 0.f;
        *reinterpret_cast<float2*>(yw + ((k + 1) * PW + l + 1) * 32) = make_float2(y0, y1);
        mxv = fmaxf(mxv, fmaxf(y0, y1));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) mxv = fmaxf(mxv, __shfl_xor(mxv, o, 64));
    if (pr == 0 && live) h.wm_out[(size_t)t * NS + s] = mxv;
    float ax;
    if (h.amax_true) ax = __uint_as_float(reinterpret_cast<const unsigned*>(h.amax_true)[b]);
    else {
      const int wmpb = a.TPB * NS;
      const float* wp = h.wm_prev + (size_t)b * wmpb;
      ax = 0.f;
      for (int i = pr; i < wmpb; i += 16) ax = fmaxf(ax, wp[i]);
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) ax = fmaxf(ax, __shfl_xor(ax, o, 64));
    }
    const float bound = h.g1 * ax + h.g0;
    const unsigned bbits = __float_as_uint(bound);
    if (s == 0 && tt == 0 && pr == 0 && live) h.amax_next[b] = bbits;
    float sb, inv_;
    wino_h2_scales(bbits, WT::VSHIFT, &sb, &inv_);
    const float sv = live ? sb : 0.f;
    __syncthreads();
    // ---- the next item's first NPRE columns: in flight during the input transform below (always issued: no run-time branch around loads)
    {
      const int nx = item + (int)gridDim.x < n_items ? item + (int)gridDim.x : item;
      const int g2 = __builtin_amdgcn_readfirstlane(nx / NS), s2 = __builtin_amdgcn_readfirstlane(nx - (nx / NS) * NS);
#pragma unroll
      for (int nu = 0; nu < NPRE; nu++)
#pragma unroll
        for (int xi = 0; xi < AL; xi++) m[nu][xi] = h2_ldf4<MAUX>(mr, m_lane, m_soff(g2, s2, xi * AL + nu));
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- input transform of the next block, one output row at a time (column pass recomputed per row from LDS)
    const unsigned v_base = (unsigned)((((g >> 3) * h.npos) * NS + s)) * 16384u + (unsigned)(g & 7) * 2048u;
    const unsigned v_pos = (unsigned)NS * 16384u;
    const float* yr = ybase;
    asm volatile("" : "+v"(yr));
    // rows in passes of <= 3: per pass every patch column is read once (AL ds_read_b64 with immediate offsets) and gives the pass's rows
#pragma unroll
    for (int i0 = 0; i0 < AL; i0 += 3) {
      constexpr int NRMAX = 3;
      const int nr = AL - i0 < NRMAX ? AL - i0 : NRMAX;   // (constant after unrolling)
      f2c tmr[NRMAX][AL];
#pragma unroll
      for (int j = 0; j < AL; j++) {
        f2c d[AL];
#pragma unroll
        for (int ii = 0; ii < AL; ii++) {
          const float2 v = *reinterpret_cast<const float2*>(yr + (ii * PW + j) * 32);
          d[ii] = f2c{v.x, v.y};
        }
#pragma unroll
        for (int r = 0; r < NRMAX; r++)
          if (r < nr) tmr[r][j] = wino_bt_row_p<TM>(i0 + r, d);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < NRMAX; r++) {
        if (r >= nr) continue;
        const int i = i0 + r;
        f2c o[AL];
        wino_btv_p<TM>(tmr[r], o);
#pragma unroll
        for (int j = 0; j < AL; j++) {
          unsigned lo;
          const unsigned hi = wino_h2_pack(o[j].x * sv, o[j].y * sv, &lo);
          const auto s16 = __builtin_amdgcn_permlane16_swap(hi, lo, false, false);
          const unsigned e0 = s16[0], e1 = s16[1];
          const auto s32 = __builtin_amdgcn_permlane32_swap(e0, e1, false, false);
          const unsigned w0 = s32[0], w1 = s32[1];
          __builtin_amdgcn_raw_buffer_store_b32(w0, vr, v_lane, v_base + (unsigned)(i * AL + j) * v_pos, VAUX);
          __builtin_amdgcn_raw_buffer_store_b32(w1, vr, v_lane + 256u, v_base + (unsigned)(i * AL + j) * v_pos, VAUX);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();   // the LDS board is rewritten by the next item's epilogue
  }
}

// U2 (conv_wino_h2.hpp: [pos][C/32][piece][Ntot][32]) -> U2c with the chained column order (host, at commit)
static void wino_build_u2c(const std::vector<_Float16>& u2, std::vector<_Float16>& u2c, int npos, int Ntot, int C, int Cout_p) {
  const int NC = C / 32, NT = Ntot / 256;
  u2c.assign(u2.size(), (_Float16)0.f);
  for (int pos = 0; pos < npos; pos++)
    for (int kc = 0; kc < NC; kc++)
      for (int nn = 0; nn < Ntot; nn++) {
        const int s = nn >> 6, j2 = nn & 63, ch = 32 * s + (j2 >> 1), br = j2 & 1, n = br * Cout_p + ch;
        for (int p = 0; p < 2; p++) {
          const _Float16* src = &u2[((((size_t)pos * NC + kc) * 2 + p) * Ntot + n) * 32];
          _Float16* dst = &u2c[((((size_t)pos * NC + kc) * NT + (nn >> 8)) * 256 + (nn & 255)) * 64 + p * 32];
          for (int k = 0; k < 32; k++) dst[k] = src[k];
        }
      }
}

// shapes the chained form takes: whole boards per 16-row group, 256-column GEMM tiles, a wave of the block-0 input transform inside one tile
static inline bool wino_h2c_ok(int H, int W, int tm, int Kp) {
  const int tpb = ceil_div(H, tm) * ceil_div(W, tm);
  const int nk = Kp / 32;
  return Kp % 128 == 0 && 16 % tpb == 0 && (nk == 4 || nk == 8 || nk == 12 || nk == 16) &&
         (size_t)(16 / tpb) * (tm * ceil_div(H, tm) + 2) * (tm * ceil_div(W, tm) + 2) * 128 <= 80 * 1024;   // the haloed LDS boards of a group, two workgroups per CU
}

// one stage of the chained tower for a chunk of boards (geometry as wino_h2_launch sets it)
static void wino_h2c_geometry(WinoH2Args& h) {
  WinoArgs& a = h.w;
  const int tm = h.tm == 5 ? 5 : 4;
  h.tm = tm; h.npos = (tm + 2) * (tm + 2);
  a.nty = ceil_div(a.H, tm); a.ntx = ceil_div(a.W, tm); a.TPB = a.nty * a.ntx; a.T = a.B * a.TPB;
  h.rsh = 7; h.rmask = 127; h.rA = (unsigned)h.npos * 128u; h.rB = 128u;
  a.n_mtiles = ceil_div(a.T, 128); a.n_ntiles = ceil_div(a.Ntot, 128);
  h.in_swap = 1; h.cform = 1; h.wm_per_board = 0;
}
static void wino_h2c_in(agz_ctx* ctx, WinoH2Args& h, hipStream_t st) {     // block 0: x -> V2c (exact range: h.amax_in, fuse_prev = 0)
  wino_h2c_geometry(h);
  h.fuse_prev = 0; h.amax_self = const_cast<unsigned*>(h.amax_in);
  ProfScopeOn ps(ctx, AGZ_PROF_WINO_IN, st == ctx->stream);
  const size_t n_in = (size_t)h.w.T * (h.w.C / 2);
  if (h.tm == 5) hipLaunchKernelGGL(wino_in_h2_kernel<5>, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, st, h);
  else hipLaunchKernelGGL(wino_in_h2_kernel<4>, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, st, h);
}
constexpr int WINO_H2_GEMM_DEFAULT = 1;   // 1: wino_gemm_h2g_kernel (three workgroups per CU), 2: wino_gemm_h2p_kernel (persistent)
static void wino_h2c_gemm(agz_ctx* ctx, WinoH2Args& h, hipStream_t st) {
  wino_h2c_geometry(h);
  ProfScopeOn ps(ctx, AGZ_PROF_WINO_GEMM, st == ctx->stream);
  const dim3 g(h.npos * h.w.n_mtiles * (h.w.Ntot >> 8));
  // (the DMA form addresses V and U2c through buffer descriptors: 31-bit byte offsets)
  const bool dma_ok = wino_h2_rows(h.npos, (size_t)h.w.T) * h.w.C * 4 < ((size_t)1 << 31) && (size_t)h.npos * h.w.C * h.w.Ntot * 4 < ((size_t)1 << 31);
  // persistent form (wino_gemm_h2p_kernel; gemm_variant 2, AGZ_WINO_H2_GEMM=2): K = 256, whole 128-column slabs, one workgroup per CU
  const int variant_all = h.gemm_variant > 0 ? h.gemm_variant : WINO_H2_GEMM_DEFAULT;   // (the environment switch is resolved by the caller: net.hip)
  const int variant = variant_all & 15, mode = (variant_all >> 4) & 3;   // (mode: agz_debug.h decomposition runs — results are then NOT valid)
  const int n_slabs = h.w.Ntot >> 7;
  if (variant == 2 && dma_ok && (h.w.C >> 5) == 8 && h.w.Ntot % 256 == 0 && ctx->num_cus / 8 >= n_slabs) {
    const dim3 gp((unsigned)(8 * ((ctx->num_cus / 8) / n_slabs) * n_slabs));
    switch (mode) {
      case 1: hipLaunchKernelGGL(wino_gemm_h2p_kernel<1>, gp, dim3(256), 0, st, h); break;
      case 2: hipLaunchKernelGGL(wino_gemm_h2p_kernel<2>, gp, dim3(256), 0, st, h); break;
      case 3: hipLaunchKernelGGL(wino_gemm_h2p_kernel<3>, gp, dim3(256), 0, st, h); break;
      default: hipLaunchKernelGGL(wino_gemm_h2p_kernel<0>, gp, dim3(256), 0, st, h); break;
    }
    return;
  }
  if (dma_ok && h.temporal_stores && (h.w.C >> 5) == 8) { hipLaunchKernelGGL((wino_gemm_h2g_kernel<8, false>), g, dim3(256), 0, st, h); return; }   // (A/B: round 4's stores)
  if (dma_ok) {
    switch (h.w.C >> 5) {
      case 4: hipLaunchKernelGGL((wino_gemm_h2g_kernel<4>), g, dim3(256), 0, st, h); return;
      case 8: hipLaunchKernelGGL((wino_gemm_h2g_kernel<8>), g, dim3(256), 0, st, h); return;
      case 12: hipLaunchKernelGGL((wino_gemm_h2g_kernel<12>), g, dim3(256), 0, st, h); return;
      default: hipLaunchKernelGGL((wino_gemm_h2g_kernel<16>), g, dim3(256), 0, st, h); return;
    }
  }
  switch (h.w.C >> 5) {
    case 4: hipLaunchKernelGGL((wino_gemm_h2c_kernel<4, 2>), g, dim3(256), 0, st, h); break;
    case 8: hipLaunchKernelGGL((wino_gemm_h2c_kernel<8, 2>), g, dim3(256), 0, st, h); break;
    case 12: hipLaunchKernelGGL((wino_gemm_h2c_kernel<12, 2>), g, dim3(256), 0, st, h); break;
    default: hipLaunchKernelGGL((wino_gemm_h2c_kernel<16, 2>), g, dim3(256), 0, st, h); break;
  }
}
// variant 4: the pipelined kernel (default); 1: the plain one (A/B hook; also the last block, whose y goes to HBM, and tensors past 2 GB)
static void wino_h2c_oi(agz_ctx* ctx, WinoH2Args& h, bool last, hipStream_t st, int variant = 4) {
  wino_h2c_geometry(h);
  ProfScopeOn ps(ctx, AGZ_PROF_WINO_OUT, st == ctx->stream);
  const WinoArgs& a = h.w;
  const dim3 g((unsigned)ceil_div(a.T, 16), (unsigned)(a.C >> 5));
  const size_t shm = last ? 0 : (size_t)(16 / a.TPB) * a.H * a.W * 32 * sizeof(float);
  // (the pipelined kernel addresses M and V through one buffer descriptor each: 31-bit byte offsets)
  const bool fits31 = wino_h2_rows(h.npos, (size_t)a.T) * a.Ntot * 4 < ((size_t)1 << 31);
  // up to 80 KB of dynamic LDS (wino_h2c_ok): a function attribute of the CURRENT DEVICE — kept per context (several contexts / devices in
  // one process: agz_comm_init_all), its result checked; where it cannot be set the plain out->in kernel below runs instead
  if (variant == 4 && !last && fits31 && !(ctx->func_attr_state & 2u)) {
    const bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_oip_h2c_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess &&
                    hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_oip_h2c_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) == hipSuccess;
    if (!ok) (void)hipGetLastError();
    ctx->func_attr_state |= 2u | (ok ? 1u : 0u);
  }
  if (variant == 4 && !last && fits31 && (ctx->func_attr_state & 1u)) {   // persistent, software-pipelined
    const size_t shp = (size_t)(16 / a.TPB) * (h.tm * a.nty + 2) * (h.tm * a.ntx + 2) * 32 * sizeof(float);
    const int items = ceil_div(a.T, 16) * (a.C >> 5);
    const dim3 gp((unsigned)std::min(items, 2 * ctx->num_cus));
    if (h.tm == 5 && h.temporal_stores) {   // A/B instance (F(5x5,3x3) shapes only): round 4's stores
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wino_oip_h2c_kernel<5, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
      hipLaunchKernelGGL((wino_oip_h2c_kernel<5, false>), gp, dim3(256), shp, st, h);
      return;
    }
    if (h.tm == 5) hipLaunchKernelGGL((wino_oip_h2c_kernel<5>), gp, dim3(256), shp, st, h);
    else hipLaunchKernelGGL((wino_oip_h2c_kernel<4>), gp, dim3(256), shp, st, h);
    return;
  }
  if (h.tm == 5) {
    if (last) hipLaunchKernelGGL((wino_oi_h2c_kernel<5, true>), g, dim3(256), shm, st, h);
    else hipLaunchKernelGGL((wino_oi_h2c_kernel<5, false>), g, dim3(256), shm, st, h);
  } else {
    if (last) hipLaunchKernelGGL((wino_oi_h2c_kernel<4, true>), g, dim3(256), shm, st, h);
    else hipLaunchKernelGGL((wino_oi_h2c_kernel<4, false>), g, dim3(256), shm, st, h);
  }
}
