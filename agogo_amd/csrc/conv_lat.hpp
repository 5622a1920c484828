// Small-batch ("latency regime") dual block: tournament Agent.Search evaluates ONE board per simulation (agent.go:77-80,
// dualnet/meta.go:168-190), so a 19x19 / K=256 layer is a 361 x 512 x 2304 GEMM over a few MB of weights that are touched for the first
// time in every evaluation — bound by how fast weights and activations get INTO the CUs and by launch / dependency latency, not by
// tiles.  Round 2 ran it as a 9-way split-K fp32-MFMA convolution plus a reduction kernel (22.7 us per layer).  Here: ONE launch per
// layer with no partial sums in memory:
//   * workgroup = (8 output channels x both branches = 16 GEMM columns) x (a slot of 48-pixel row groups); NW = C/32 waves, wave w
//     owns input channels [32 w, 32 w + 32) — the K split is INSIDE the workgroup;
//   * every wave issues the loads of its whole weight slice up front (9 taps x pieces x one 16-byte MFMA B fragment, buffer loads with
//     one lane offset and scalar offsets), kept in registers for all row groups of the workgroup; every weight byte is read by exactly
//     the workgroups of one column tile, which share an XCD's L2;
//   * the wave's activation slice (the pixels of the row group plus its 3x3 halo, 32 channels) is split once on its way into LDS
//     ([piece][pixel][32], wave-private); the nine taps read their A fragments from there (no barrier: a wave reads what it wrote);
//   * v_mfma_f32_16x16x32 over three interleaved 16-row tiles; the NW partial tiles are summed through LDS in wave order (deterministic,
//     the same for every batch size of the regime) and the BN / ReLU / dual-add epilogue is applied by the same workgroup.
// The library's kernel is the fp16x2 form (conv3x3_lat_h2_kernel below: 3 products per tap, 4.7 MB of weights per layer, 8.8 us per
// layer); the bf16x3 form of round 3's first version (6 products, 7 MB, 10.2 us; its phases are in profiles/r03/latency_tower.md)
// is compiled only into scripts/probes/lat_probe.hip.
#pragma once
// (included by net.hip INSIDE namespace agz, after conv_x3.hpp)

typedef float f32x4_lat __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_lat __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lat_lds_ptr_t;
constexpr int LAT_ROWS = 48;         // pixels per row group (3 MFMA row tiles)
constexpr int LAT_NPIX = 100;        // pixels of a row group incl. halo: 48 + up to 3 board-row seams * 2 + 2 * (Wp + 1) <= 98 at Wp = 21
constexpr int LAT_SLOTS = 8;         // row-group slots (grid.y)

#ifdef LAT_PROBE   // the bf16x3 form of round 3's first version: kept for scripts/probes/lat_probe.hip (phase stamps), not part of the library
struct LatArgs {
  const float* x;            // [B][Hp][Wp][C] padded NHWC fp32 (PRE = false: split in the kernel)
  const unsigned short* x3;  // [B][3][Hp*Wp][C] the same activations as three bf16 pieces, written by the previous layer (PRE = true)
  const unsigned short* w3;  // [C/16][9][3][Ntot][16] bf16 pieces, columns in block-tile order (tile*128 + branch*64 + c%64)
  const void* ep;            // float4 {sa,ta,sb,tb} [HW][Cout_p]
  float* y;                  // [B][Hp][Wp][Cout_p]
  unsigned short* y3;        // [B][3][Hp*Wp][Cout_p] pieces of y for the next layer (nullptr: not written)
  int B, H, W, Hp, Wp, C, Cout_p, Ntot;
  int groups_per_board;      // ceil(HW / 48)
#ifdef LAT_PROBE
  long long* dbg;            // scripts/probes/lat_probe.hip: s_memrealtime stamps of one wave
#endif
};
#ifdef LAT_PROBE
#define LAT_STAMP(i, waitasm) do { if (a.dbg) { asm volatile(waitasm ::: "memory"); if (blockIdx.x == LAT_PROBE_X && blockIdx.y == 0 && w == LAT_PROBE_W) { long long t_ = __builtin_amdgcn_s_memrealtime(); if (lane == 0) a.dbg[i] = t_; } } } while (0)
#else
#define LAT_STAMP(i, waitasm) do { } while (0)
#endif

template <int NW, bool PRE>
__global__ __launch_bounds__(NW * 64) void conv3x3_lat_x3_kernel(LatArgs a) {
  constexpr int PIECE = LAT_NPIX * 64;                 // bytes of one piece image of one wave (32 bf16 per pixel)
  constexpr int WAVE_LDS = 3 * PIECE;                  // 19,200 B
  __shared__ __attribute__((aligned(16))) unsigned char lds[NW * WAVE_LDS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = blockIdx.x;                           // column tile: channels ct*8 .. ct*8+7, both branches
  const int HW = a.H * a.W, HpWp = a.Hp * a.Wp;
  const int r16 = lane & 15, q = lane >> 4;

  LAT_STAMP(0, "");
  unsigned char* my = lds + w * WAVE_LDS;
  const int n_groups = a.B * a.groups_per_board;
  const int sub = lane >> 3, c4 = (lane & 7) * 4;     // fp32 staging: lane = (pixel l/8 of 8, 4 channels (l%8)*4 ..) per instruction
  constexpr int NIT = PRE ? 1 : (LAT_NPIX + 7) / 8;
  constexpr int NE = (LAT_ROWS * 8 + NW * 64 - 1) / (NW * 64);      // epilogue items per thread
  float4 v[NIT];
  float4 E[NE];
  // the epilogue parameters of a group's outputs
  auto issue_params = [&](int grp) {
    const int b = grp / a.groups_per_board, gi = grp - b * a.groups_per_board;
    const int p0 = gi * LAT_ROWS;
#pragma unroll
    for (int k = 0; k < NE; k++) {
      const int o = tid + k * NW * 64;
      int p = p0 + (o >> 3);
      p = p < HW ? p : HW - 1;
      E[k] = reinterpret_cast<const float4*>(a.ep)[(size_t)p * a.Cout_p + ct * 8 + (o & 7)];
    }
  };
  // the wave's activation slice of a row group (its pixels plus the 3x3 halo, 32 channels).  Buffer loads: one lane-offset register
  // for the whole batch (global_load needs an address pair per load — registers the allocator then recycles by WAITING for early
  // loads before the weight loads are out), and a window that runs past the last board reads zeros instead of faulting.
  auto issue_group = [&](int grp) {
    const int b = grp / a.groups_per_board, gi = grp - b * a.groups_per_board;
    const int p0 = gi * LAT_ROWS;
    const int pix0 = p0 + 2 * (p0 / a.W);
    if constexpr (PRE) {
      // pieces written by the previous layer: DMA straight into the LDS image, 16 pixels x 64 B per instruction, no registers, no VALU
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.x3), 0, a.B * 3 * HpWp * a.C * 2, 0x00020000);
      const unsigned vo = (unsigned)((((b * 3) * HpWp + pix0 + (lane >> 2)) * a.C + 32 * w + (lane & 3) * 8) * 2);
      const unsigned pl = (unsigned)(HpWp * a.C * 2), step = (unsigned)(a.C * 32);
#pragma unroll
      for (int p = 0; p < 3; p++) {
#pragma unroll
        for (int it = 0; it < LAT_NPIX / 16; it++)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lat_lds_ptr_t)(my + p * PIECE + it * 1024), 16, vo + p * pl + it * step, 0, 0, 0);
        if (lane < (LAT_NPIX % 16) * 4)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rx, (lat_lds_ptr_t)(my + p * PIECE + (LAT_NPIX / 16) * 1024), 16, vo + p * pl + (LAT_NPIX / 16) * step, 0, 0, 0);
      }
    } else {
      const float* xb = a.x + ((size_t)b * HpWp + pix0) * a.C + 32 * w;
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, ((a.B - b) * HpWp - pix0) * a.C * 4 - 128 * w, 0x00020000);
      const unsigned vo = (unsigned)(sub * a.C + c4) * 4u, step = (unsigned)a.C * 32u;
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const u32x4_lat r = __builtin_amdgcn_raw_buffer_load_b128(rx, vo + it * step, 0, 0);
        v[it] = make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
      }
    }
  };
  // Issue order (loads return in order): parameters, the first group's activations, [workgroup barrier: every wave's activation loads
  // are queued before any wave's weight loads], the weights.  The activations are consumed while the weights are still in flight.
  issue_params(blockIdx.y);                          // (grid.y <= n_groups: conv_lat_launch)
  issue_group(blockIdx.y);
  __builtin_amdgcn_sched_barrier(0);                 // (the compiler otherwise hoists the weight loads above these)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  // ---- the wave's weight slice, all of it: B fragment of (tap, piece): column j = lane % 16 (branch j / 8, channel ct*8 + j % 8),
  // k octet q of the wave's 32 channels = chunk 2w + q/2, elements (q%2)*8 .. +7
  bf16x8_t Bf[9][3];
  {
    const int j = r16, c = ct * 8 + (j & 7);
    const int n = (c >> 6) * 128 + (j >> 3) * 64 + (c & 63);
    const int cc = 2 * w + (q >> 1);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(a.w3), 0, (a.C / 16) * 27 * a.Ntot * 32, 0x00020000);
    const unsigned vo = (unsigned)((cc * 27 * a.Ntot + n) * 16 + (q & 1) * 8) * 2u;
    const unsigned pstride = (unsigned)a.Ntot * 32u;    // bytes between (tap, piece) planes: a scalar offset per load
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
      for (int p = 0; p < 3; p++) {
        const u32x4_lat r = __builtin_amdgcn_raw_buffer_load_b128(rw, vo, (t * 3 + p) * pstride, 0);
        Bf[t][p] = __builtin_bit_cast(bf16x8_t, r);
      }
  }

  // (a lambda instantiated twice, not a loop entered with the weights in flight: the wait counts the compiler inserts at a loop
  // header must hold for every way of reaching it, so a shared body would wait for the WEIGHTS before it touches the first group)
  auto process = [&](const int grp, auto first_c) __attribute__((always_inline)) {
    constexpr bool first = decltype(first_c)::value;
    const int b = grp / a.groups_per_board, gi = grp - b * a.groups_per_board;
    const int p0 = gi * LAT_ROWS;                       // first pixel (row-major interior index) of the group
    // pixel window of the group in the padded board: centre of p = p + 2 (p / W) + Wp + 1
    const int pix0 = p0 + 2 * (p0 / a.W);               // = centre(p0) - Wp - 1
    LAT_STAMP(1, "s_waitcnt vmcnt(27)");              // (probe: activations + parameters arrived, weights in flight)
    if constexpr (PRE) {
      // the DMA writes LDS behind the compiler's back: wait for it by hand (first group: the 27 weight loads stay in flight)
      if constexpr (first) asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      // ---- split into the three bf16 pieces, once per element, on the way into the wave's LDS image
#pragma unroll
      for (int it = 0; it < NIT; it++) {
        const int px = it * 8 + sub;
        unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
        x3_split(v[it].x, h0, m0, l0); x3_split(v[it].y, h1, m1, l1); x3_split(v[it].z, h2, m2, l2); x3_split(v[it].w, h3, m3, l3);
        if (px < LAT_NPIX) {
          unsigned char* dst = my + px * 64 + c4 * 2;
          *reinterpret_cast<uint2*>(dst) = make_uint2(x3_pack(h0, h1), x3_pack(h2, h3));
          *reinterpret_cast<uint2*>(dst + PIECE) = make_uint2(x3_pack(m0, m1), x3_pack(m2, m3));
          *reinterpret_cast<uint2*>(dst + 2 * PIECE) = make_uint2(x3_pack(l0, l1), x3_pack(l2, l3));
        }
      }
      // (wave-private LDS: the wave's own writes are visible to its own reads once they have completed)
      __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0)
    }
    __builtin_amdgcn_wave_barrier();
    LAT_STAMP(2, "");
    LAT_STAMP(3, "s_waitcnt vmcnt(0)");

    // ---- 3 row tiles x 9 taps x 6 piece products
    f32x4_lat acc[3];
    unsigned loc[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      acc[i] = f32x4_lat{0.f, 0.f, 0.f, 0.f};
      int p = p0 + i * 16 + r16;
      if (p >= HW) p = HW - 1;                          // (rows past the board: computed on a valid pixel, never stored)
      loc[i] = (unsigned)((p + 2 * (p / a.W) + a.Wp + 1 - pix0) * 64 + q * 16);
    }
#pragma unroll
    for (int t = 0; t < 9; t++) {
      const int toff = ((t / 3 - 1) * a.Wp + (t % 3 - 1)) * 64;
      bf16x8_t A_[3][3];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const unsigned char* src = my + (int)loc[i] + toff;
#pragma unroll
        for (int p = 0; p < 3; p++) A_[i][p] = *reinterpret_cast<const bf16x8_t*>(src + p * PIECE);
      }
      // small terms first (conv_x3.hpp): lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi; the three row tiles interleaved so that
      // consecutive MFMAs never wait on each other's accumulator
#pragma unroll
      for (int k = 0; k < 6; k++) {
        const int pa = k == 0 ? 2 : (k == 2 || k == 3) ? 1 : 0, pb = k == 1 ? 2 : (k == 2 || k == 4) ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 3; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A_[i][pa], Bf[t][pb], acc[i], 0, 0, 0);
      }
    }

    LAT_STAMP(4, "");
    // ---- sum the NW channel slices through LDS (wave order), epilogue
    __syncthreads();                                    // every wave is done with its activation image
    float* red = reinterpret_cast<float*>(lds);         // [NW][48 rows][16 columns]
    {
      // C/D layout of 16x16: column = lane % 16, rows 4 * (lane / 16) + reg
#pragma unroll
      for (int i = 0; i < 3; i++)
#pragma unroll
        for (int r = 0; r < 4; r++) red[((size_t)w * LAT_ROWS + i * 16 + 4 * q + r) * 16 + r16] = acc[i][r];
    }
    __syncthreads();
    LAT_STAMP(5, "");
#pragma unroll
    for (int k = 0; k < NE; k++) {
      const int o = tid + k * NW * 64;
      const int row = o >> 3, ch = o & 7;
      const int p = p0 + row;
      if (o < LAT_ROWS * 8 && p < HW) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int u = 0; u < NW; u++) {
          sa += red[((size_t)u * LAT_ROWS + row) * 16 + ch];
          sb += red[((size_t)u * LAT_ROWS + row) * 16 + 8 + ch];
        }
        const int c = ct * 8 + ch;
        const float4 e = E[k];
        float va = sa * e.x + e.y, vb = sb * e.z + e.w;
        va = va > 0.f ? va : 0.f;
        vb = vb > 0.f ? vb : 0.f;
        const float s_ = va + vb;
        const float out = s_ > 0.f ? s_ : 0.f;
        const int h = p / a.W, ww = p - h * a.W;
        const size_t pix = (size_t)(h + 1) * a.Wp + (ww + 1);
        a.y[((size_t)b * HpWp + pix) * a.Cout_p + c] = out;
        if (a.y3) {                                     // the next layer's operand, split here once instead of by 32 column tiles
          unsigned hh, mm, ll;
          x3_split(out, hh, mm, ll);
          unsigned short* d3 = a.y3 + ((size_t)(b * 3) * HpWp + pix) * a.Cout_p + c;
          const size_t pl = (size_t)HpWp * a.Cout_p;
          d3[0] = (unsigned short)(hh >> 16); d3[pl] = (unsigned short)(mm >> 16); d3[2 * pl] = (unsigned short)(ll >> 16);
        }
      }
    }
    LAT_STAMP(6, "s_waitcnt vmcnt(0)");
  };
  process((int)blockIdx.y, std::integral_constant<bool, true>{});
  for (int grp = blockIdx.y + LAT_SLOTS; grp < n_groups; grp += LAT_SLOTS) {   // (batches of more than one board)
    __syncthreads();                                    // the reduction buffer is the next group's activation image
    issue_params(grp);
    issue_group(grp);
    process(grp, std::integral_constant<bool, false>{});
  }
}

#endif  // LAT_PROBE

// ---- the same layer with fp16x2 products -----------------------------------------------------------------------------------
// Half the matrix instructions (hi hi, hi lo, lo hi of v_mfma_f32_16x16x32_f16 instead of six bf16 products) and two thirds of the
// weight bytes (4 B per weight instead of 6), which is what the layer is bound by (per-CU fill rate, profiles/r03/latency_tower.md).
// Ranges as in conv_wino_h2.hpp: the weight image is equilibrated at commit — row ci divided by t_in[ci] = 2^floor(log2 max|w[.][ci][.]|),
// column n multiplied by su[n] so that its maximum lands in [2^13, 2^14) — and the activations are multiplied by t_in[ci] and by the
// BOARD's power of two sb = 2^(13 - E(max |x t_in|)) on their way into LDS (so a board's result does not depend on its batch
// neighbours).  That maximum comes from the previous layer: every workgroup stores the maximum of (its outputs x next layer's t_in),
// the consumer's waves reduce their board's words themselves (n_in_words each; layer 0: one word from board_amax_kernel).
struct LatH2Args {
  const float* x;            // [B][Hp][Wp][C] padded NHWC fp32
  const _Float16* w2;        // [C/32][9][2][Ntot][32] fp16 hi / lo, equilibrated, columns in block-tile order (tile*128 + branch*64 + c%64)
  const float* t_in;         // [C]
  const float* col_unscale;  // [Ntot] 1 / su[n]
  const float* t_next;       // [Cout_p] the next layer's t_in (nullptr: no range words written)
  const float* wmax_in;      // [B][n_in_words] maxima of |x t_in| (>= 0)
  float* wmax_out;           // [B][groups_per_board * Cout_p / 8]
  const void* ep;            // float4 {sa,ta,sb,tb} [HW][Cout_p]
  float* y;                  // [B][Hp][Wp][Cout_p]
  int B, H, W, Hp, Wp, C, Cout_p, Ntot;
  int groups_per_board, n_in_words;
};
typedef _Float16 lat_f16x8_t __attribute__((ext_vector_type(8)));

template <int NW>
__global__ __launch_bounds__(NW * 64) void conv3x3_lat_h2_kernel(LatH2Args a) {
  constexpr int PIECE = LAT_NPIX * 64;                 // bytes of one piece image of one wave (32 fp16 per pixel)
  constexpr int WAVE_LDS = 2 * PIECE;                  // 12,800 B
  constexpr int RED_BYTES = NW * LAT_ROWS * 16 * 4;    // the reduction buffer overlays the images
  __shared__ __attribute__((aligned(16))) unsigned char lds[(NW * WAVE_LDS > RED_BYTES ? NW * WAVE_LDS : RED_BYTES) + 64];
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = blockIdx.x;
  const int HW = a.H * a.W, HpWp = a.Hp * a.Wp;
  const int r16 = lane & 15, q = lane >> 4;
  unsigned char* my = lds + w * WAVE_LDS;
  float* wgmax = reinterpret_cast<float*>(lds + (NW * WAVE_LDS > RED_BYTES ? NW * WAVE_LDS : RED_BYTES));   // [NW] (+ pad)
  const int n_groups = a.B * a.groups_per_board;
  const int sub = lane >> 3, c4 = (lane & 7) * 4;
  constexpr int NIT = (LAT_NPIX + 7) / 8;
  constexpr int NE = (LAT_ROWS * 8 + NW * 64 - 1) / (NW * 64);
  constexpr int NWRD = 4;                              // range words per lane (n_in_words <= 256)
  float4 v[NIT];
  float4 E[NE];
  float wr[NWRD];
  const float4 t4 = *reinterpret_cast<const float4*>(a.t_in + 32 * w + c4);

  auto issue_params = [&](int grp) {
    const int b = grp / a.groups_per_board, gi = grp - b * a.groups_per_board;
    const int p0 = gi * LAT_ROWS;
#pragma unroll
    for (int k = 0; k < NE; k++) {
      const int o = tid + k * NW * 64;
      int p = p0 + (o >> 3);
      p = p < HW ? p : HW - 1;
      E[k] = reinterpret_cast<const float4*>(a.ep)[(size_t)p * a.Cout_p + ct * 8 + (o & 7)];
    }
  };
  auto issue_group = [&](int grp) {
    const int b = grp / a.groups_per_board, gi = grp - b * a.groups_per_board;
    const int p0 = gi * LAT_ROWS;
    const int pix0 = p0 + 2 * (p0 / a.W);
    // the board's range words first (they return first), then the activation slice
    const float* wp = a.wmax_in + (size_t)b * a.n_in_words;
#pragma unroll
    for (int k = 0; k < NWRD; k++) { const int i = lane + k * 64; wr[k] = wp[i < a.n_in_words ? i : 0]; }
    const float* xb = a.x + ((size_t)b * HpWp + pix0) * a.C + 32 * w;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, ((a.B - b) * HpWp - pix0) * a.C * 4 - 128 * w, 0x00020000);
    const unsigned vo = (unsigned)(sub * a.C + c4) * 4u, step = (unsigned)a.C * 32u;
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const u32x4_lat r = __builtin_amdgcn_raw_buffer_load_b128(rx, vo + it * step, 0, 0);
      v[it] = make_float4(__uint_as_float(r[0]), __uint_as_float(r[1]), __uint_as_float(r[2]), __uint_as_float(r[3]));
    }
  };
  issue_params(blockIdx.y);
  issue_group(blockIdx.y);
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();                      // every wave's activation loads are queued before any wave's weight loads
  __builtin_amdgcn_sched_barrier(0);

  // the wave's weight slice: B fragment of (tap, piece) = 8 k of column j: chunk w, halves q*8 .. +7
  lat_f16x8_t Bf[9][2];
  {
    const int j = r16, c = ct * 8 + (j & 7);
    const int n = (c >> 6) * 128 + (j >> 3) * 64 + (c & 63);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.w2), 0, (a.C / 32) * 18 * a.Ntot * 64, 0x00020000);
    const unsigned vo = (unsigned)((w * 18 * a.Ntot + n) * 32 + q * 8) * 2u;
    const unsigned pstride = (unsigned)a.Ntot * 64u;    // bytes between (tap, piece) planes
#pragma unroll
    for (int t = 0; t < 9; t++)
#pragma unroll
      for (int p = 0; p < 2; p++) {
        const u32x4_lat r = __builtin_amdgcn_raw_buffer_load_b128(rw, vo, (t * 2 + p) * pstride, 0);
        Bf[t][p] = __builtin_bit_cast(lat_f16x8_t, r);
      }
  }

  auto process = [&](const int grp) __attribute__((always_inline)) {
    const int b = grp / a.groups_per_board, gi = grp - b * a.groups_per_board;
    const int p0 = gi * LAT_ROWS;
    const int pix0 = p0 + 2 * (p0 / a.W);
    // ---- the board's scale, then the split into hi / lo fp16 on the way into the wave's LDS image
    float mx = fmaxf(fmaxf(wr[0], wr[1]), fmaxf(wr[2], wr[3]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sb, inv_sb;
    h2_scales(__float_as_uint(mx), &sb, &inv_sb);
    const float4 s4 = make_float4(sb * t4.x, sb * t4.y, sb * t4.z, sb * t4.w);
#pragma unroll
    for (int it = 0; it < NIT; it++) {
      const int px = it * 8 + sub;
      const float x0 = v[it].x * s4.x, x1 = v[it].y * s4.y, x2 = v[it].z * s4.z, x3 = v[it].w * s4.w;
      const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1, h2 = (_Float16)x2, h3 = (_Float16)x3;
      const _Float16 l0 = (_Float16)(x0 - (float)h0), l1 = (_Float16)(x1 - (float)h1), l2 = (_Float16)(x2 - (float)h2), l3 = (_Float16)(x3 - (float)h3);
      if (px < LAT_NPIX) {
        unsigned char* dst = my + px * 64 + c4 * 2;
        typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
        *reinterpret_cast<f16x4_t*>(dst) = f16x4_t{h0, h1, h2, h3};
        *reinterpret_cast<f16x4_t*>(dst + PIECE) = f16x4_t{l0, l1, l2, l3};
      }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);               // lgkmcnt(0): the wave's own LDS writes are visible to its own reads
    __builtin_amdgcn_wave_barrier();

    f32x4_lat acc[3];
    unsigned loc[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
      acc[i] = f32x4_lat{0.f, 0.f, 0.f, 0.f};
      int p = p0 + i * 16 + r16;
      if (p >= HW) p = HW - 1;
      loc[i] = (unsigned)((p + 2 * (p / a.W) + a.Wp + 1 - pix0) * 64 + q * 16);
    }
#pragma unroll
    for (int t = 0; t < 9; t++) {
      const int toff = ((t / 3 - 1) * a.Wp + (t % 3 - 1)) * 64;
      lat_f16x8_t A_[3][2];
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const unsigned char* src = my + (int)loc[i] + toff;
#pragma unroll
        for (int p = 0; p < 2; p++) A_[i][p] = *reinterpret_cast<const lat_f16x8_t*>(src + p * PIECE);
      }
#pragma unroll
      for (int k = 0; k < 3; k++) {                    // lo*hi, hi*lo, hi*hi; the three row tiles interleaved
        const int pa = k == 0 ? 1 : 0, pb = k == 1 ? 1 : 0;
#pragma unroll
        for (int i = 0; i < 3; i++) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A_[i][pa], Bf[t][pb], acc[i], 0, 0, 0);
      }
    }

    __syncthreads();
    float* red = reinterpret_cast<float*>(lds);
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int r = 0; r < 4; r++) red[((size_t)w * LAT_ROWS + i * 16 + 4 * q + r) * 16 + r16] = acc[i][r];
    __syncthreads();
    float omx = 0.f;
#pragma unroll
    for (int k = 0; k < NE; k++) {
      const int o = tid + k * NW * 64;
      const int row = o >> 3, ch = o & 7;
      const int p = p0 + row;
      if (o < LAT_ROWS * 8 && p < HW) {
        float sa = 0.f, sbr = 0.f;
#pragma unroll
        for (int u = 0; u < NW; u++) {
          sa += red[((size_t)u * LAT_ROWS + row) * 16 + ch];
          sbr += red[((size_t)u * LAT_ROWS + row) * 16 + 8 + ch];
        }
        const int c = ct * 8 + ch;
        const int na = (c >> 6) * 128 + (c & 63);       // column of branch a in the image (branch b: + 64)
        const float4 e = E[k];
        float va = (sa * (inv_sb * a.col_unscale[na])) * e.x + e.y, vb = (sbr * (inv_sb * a.col_unscale[na + 64])) * e.z + e.w;
        va = va > 0.f ? va : 0.f;
        vb = vb > 0.f ? vb : 0.f;
        const float s_ = va + vb;
        const float out = s_ > 0.f ? s_ : 0.f;
        const int h = p / a.W, ww = p - h * a.W;
        a.y[((size_t)b * HpWp + (size_t)(h + 1) * a.Wp + (ww + 1)) * a.Cout_p + c] = out;
        if (a.t_next) omx = fmaxf(omx, out * a.t_next[c]);
      }
    }
    if (a.wmax_out) {                                   // the workgroup's range word for the next layer
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) omx = fmaxf(omx, __shfl_xor(omx, o, 64));
      if (lane == 0) wgmax[w] = omx;
      __syncthreads();
      if (tid == 0) {
        float m = wgmax[0];
#pragma unroll
        for (int u = 1; u < NW; u++) m = fmaxf(m, wgmax[u]);
        a.wmax_out[(size_t)b * (a.groups_per_board * (a.Cout_p / 8)) + gi * (a.Cout_p / 8) + ct] = m;
      }
    }
  };
  process((int)blockIdx.y);
  for (int grp = blockIdx.y + LAT_SLOTS; grp < n_groups; grp += LAT_SLOTS) {
    __syncthreads();
    issue_params(grp);
    issue_group(grp);
    process(grp);
  }
}

// range words of the FIRST layer's input: max |x t_in| over a slice of the board per workgroup, `parts` words per board (no atomics,
// nothing to clear; one workgroup per board takes 37 us for a 19x19 x 256 board)
__global__ __launch_bounds__(256) void lat_board_words_kernel(const float* __restrict__ x, const float* __restrict__ t, float* __restrict__ words,
                                                              int HW, int W, int Wp, int HpWp, int C, int parts) {
  __shared__ float red[4];
  const int b = blockIdx.x / parts, part = blockIdx.x - b * parts, tid = threadIdx.x;
  const int c4 = C >> 2;
  const int n = HW * c4, lo = (int)((long)n * part / parts), hi = (int)((long)n * (part + 1) / parts);
  float m = 0.f;
  for (int i = lo + tid; i < hi; i += 256) {
    const int p = i / c4, c = (i - p * c4) << 2;
    const int h = p / W, w = p - h * W;
    const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)b * HpWp + (h + 1) * Wp + (w + 1)) * C + c);
    const float4 tt = *reinterpret_cast<const float4*>(t + c);
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x * tt.x), fabsf(v.y * tt.y)), fmaxf(fabsf(v.z * tt.z), fabsf(v.w * tt.w))));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) words[(size_t)b * parts + part] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---- latency regime, input layer in ONE launch: convolution of the (<= 32) input planes, epilogue (BatchNorm + ReLU) and the range
// words of the first dual layer's input.  Round 3 ran a split-K MFMA convolution (9 splits), its reduction + epilogue and the range
// reduction as three launches (5.9 + 4.8 + 4.4 us and two more ~2.8 us boundaries per simulation).  53 MFLOP at 19 x 19 / K = 256 need
// no matrix pipe: a workgroup = 8 consecutive pixels x 64 output channels, thread = one channel x two pixels, the 8 x 9 x 32 input
// window and the 288 x 64 weights (a [tap][plane][channel] image) in LDS; one fp32 FMA chain per output in (tap, plane) order, so a
// board's result does not depend on what else is in the batch.  words[b][group][channel group] = max |y t| of the workgroup's outputs.
struct LatInArgs {
  const float* x; const float* w; const float2* ep; const float* t; float* y; float* words;
  int B, H, W, Hp, Wp, Cout_p, groups_per_board;
};
constexpr int LAT_IN_PIX = 8;
__global__ __launch_bounds__(256) void lat_input_kernel(LatInArgs a) {
  // wt[tap][plane][channel]: the 288 x 64 weights of the workgroup's channels are DMA'd into LDS (16 bytes per lane: four rows of 256
  // bytes per instruction, 18 instructions per wave) while the window is filled and the epilogue terms are read: everything is a
  // first touch after the previous simulation's tower, so every DEPENDENT batch of loads costs a 2-3 us round trip — weights fetched tap
  // by tap into registers: 13.7 us; in three batches: 12.8; all 288 into registers: spills; left to the compiler: two loads per
  // s_waitcnt vmcnt(0), 47 us.
  __shared__ __attribute__((aligned(16))) float wl[288][64];
  __shared__ __attribute__((aligned(16))) float xs[LAT_IN_PIX][9][32];
  __shared__ float red[4];
  const int cg = blockIdx.x, b = blockIdx.y / a.groups_per_board, grp = blockIdx.y - b * a.groups_per_board;
  const int tid = threadIdx.x, lane = tid & 63, c = cg * 64 + lane, psub = tid >> 6;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int HW = a.H * a.W, p0 = grp * LAT_IN_PIX;
  {
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, 9 * 32 * a.Cout_p * 4, 0x00020000);
    const unsigned vo = (unsigned)(((lane >> 4) * a.Cout_p + cg * 64 + (lane & 15) * 4) * 4);
    typedef __attribute__((address_space(3))) void* lds_ptr_t;
#pragma unroll
    for (int i = 0; i < 18; i++) {
      const int row0 = wid * 72 + i * 4;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_ptr_t)&wl[row0][0], 16, vo, (unsigned)(row0 * a.Cout_p) * 4u, 0, 0);
    }
  }
  const int pa = 2 * psub, pb = pa + 1;
  const float tc = a.t[c];
  float2 ep2[2];
#pragma unroll
  for (int j = 0; j < 2; j++) ep2[j] = a.ep[(size_t)min(p0 + pa + j, HW - 1) * a.Cout_p + c];
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const int e = tid + 256 * k;                    // 8 pixels x 9 taps x 8 float4
    if (e < LAT_IN_PIX * 72) {
      const int px = e / 72, r = e - px * 72, tap = r >> 3, c4 = (r & 7) << 2;
      const int p = p0 + px;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p < HW) {
        const int h = p / a.W, w = p - h * a.W, ky = tap / 3, kx = tap - ky * 3;
        v = *reinterpret_cast<const float4*>(a.x + (((size_t)b * a.Hp + h + ky) * a.Wp + w + kx) * 32 + c4);   // (padded input: pixel (h, w) sits at (h + 1, w + 1))
      }
      *reinterpret_cast<float4*>(&xs[px][tap][c4]) = v;
    }
  }
  __syncthreads();                                  // (its fence waits vmcnt(0): this wave's DMA has landed; then every wave's)
  float acc0 = 0.f, acc1 = 0.f;
#pragma unroll 3
  for (int tap = 0; tap < 9; tap++) {
#pragma unroll
    for (int q = 0; q < 8; q++) {
      const float4 x0 = *reinterpret_cast<const float4*>(&xs[pa][tap][4 * q]);
      const float4 x1 = *reinterpret_cast<const float4*>(&xs[pb][tap][4 * q]);
      const float w0 = wl[tap * 32 + 4 * q][lane], w1 = wl[tap * 32 + 4 * q + 1][lane], w2 = wl[tap * 32 + 4 * q + 2][lane], w3 = wl[tap * 32 + 4 * q + 3][lane];
      acc0 = fmaf(x0.x, w0, acc0); acc0 = fmaf(x0.y, w1, acc0); acc0 = fmaf(x0.z, w2, acc0); acc0 = fmaf(x0.w, w3, acc0);
      acc1 = fmaf(x1.x, w0, acc1); acc1 = fmaf(x1.y, w1, acc1); acc1 = fmaf(x1.z, w2, acc1); acc1 = fmaf(x1.w, w3, acc1);
    }
  }
  float m = 0.f;
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int p = p0 + pa + j;
    if (p < HW) {
      const int h = p / a.W, w = p - h * a.W;
      const float2 e = ep2[j];
      float v = (j ? acc1 : acc0) * e.x + e.y;
      v = v > 0.f ? v : 0.f;
      a.y[(((size_t)b * a.Hp + h + 1) * a.Wp + w + 1) * a.Cout_p + c] = v;
      m = fmaxf(m, fabsf(v * tc));
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) a.words[((size_t)b * a.groups_per_board + grp) * gridDim.x + cg] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

static void conv_lat_h2_launch(agz_ctx* ctx, const LatH2Args& a) {
  const dim3 grid((unsigned)(a.Cout_p / 8), (unsigned)std::min(LAT_SLOTS, a.B * a.groups_per_board));
  switch (a.C / 32) {
    case 2: hipLaunchKernelGGL((conv3x3_lat_h2_kernel<2>), grid, dim3(128), 0, ctx->stream, a); break;
    case 4: hipLaunchKernelGGL((conv3x3_lat_h2_kernel<4>), grid, dim3(256), 0, ctx->stream, a); break;
    default: hipLaunchKernelGGL((conv3x3_lat_h2_kernel<8>), grid, dim3(512), 0, ctx->stream, a); break;
  }
}

// shapes the kernel is instantiated for: C = Cout_p in {64, 128, 256} (NW = C / 32 waves; 512 would need 300 KB of LDS), boards whose
// row groups fit the window
static inline bool conv_lat_ok(int C, int Cout_p, int Wp) {
  return C == Cout_p && (C == 64 || C == 128 || C == 256) && LAT_ROWS + 2 * ((LAT_ROWS - 1) / (Wp - 2) + 1) + 2 * (Wp + 1) <= LAT_NPIX;
}

#ifdef LAT_PROBE
static void conv_lat_launch(agz_ctx* ctx, const LatArgs& a) {
  const dim3 grid((unsigned)(a.Cout_p / 8), (unsigned)std::min(LAT_SLOTS, a.B * a.groups_per_board));
  if (a.x3) {
    switch (a.C / 32) {
      case 2: hipLaunchKernelGGL((conv3x3_lat_x3_kernel<2, true>), grid, dim3(128), 0, ctx->stream, a); break;
      case 4: hipLaunchKernelGGL((conv3x3_lat_x3_kernel<4, true>), grid, dim3(256), 0, ctx->stream, a); break;
      default: hipLaunchKernelGGL((conv3x3_lat_x3_kernel<8, true>), grid, dim3(512), 0, ctx->stream, a); break;
    }
  } else {
    switch (a.C / 32) {
      case 2: hipLaunchKernelGGL((conv3x3_lat_x3_kernel<2, false>), grid, dim3(128), 0, ctx->stream, a); break;
      case 4: hipLaunchKernelGGL((conv3x3_lat_x3_kernel<4, false>), grid, dim3(256), 0, ctx->stream, a); break;
      default: hipLaunchKernelGGL((conv3x3_lat_x3_kernel<8, false>), grid, dim3(512), 0, ctx->stream, a); break;
    }
  }
}
#endif
