// agz_net: the dual network (dualnet/dual.go:50-103) as device-resident parameters + HIP launch chain.
#pragma once
#include <string>
#include <vector>

#include "common.hpp"

namespace agz {
struct Param {
  std::string name;
  std::vector<float> v;        // row-0 values (see include/agz.h)
  std::vector<int> ref_shape;  // reference tensor shape (batch-shaped for BN/bias) for the Glorot fan
  int kind;                    // 0 conv filter, 1 BN gamma/beta, 2 FC weight, 3 FC bias
};
struct BNStats { std::vector<float> mean, var; };
int conv3x3_raw(agz_ctx* ctx, const float* x, const float* w, float* y, int B, int H, int W, int Cin_p, int Cout_p);
// the same GEMM on the bf16 matrix pipe (conv_x3.hpp): w3 = split_w3() image of w; Cin_p % 16 == 0
int conv3x3_raw_x3(agz_ctx* ctx, const float* x, const unsigned short* w3, float* y, int B, int H, int W, int Cin_p, int Cout_p);
// w [9][N][Cin_p] fp32 (device) -> w3 [Cin_p/16][9][3][N][16] bf16 pieces (device); exact truncation split
int split_w3(agz_ctx* ctx, const float* w, unsigned short* w3, int N, int Cin_p);
inline size_t w3_elems(int N, int Cin_p) { return (size_t)9 * 3 * N * Cin_p; }
// the same convolution through the Winograd fp16x2 path (conv_wino_h2.hpp) with its weights transformed on the device from
// w [9][Cout_p][Cin_p] fp32: input range per board, U2 build (two small kernels), input transform, GEMMs, raw output transform.
// Cin_p % 32 == 0.  The scratch grows on demand and is owned by the caller (dual.Train: one per trainer).
struct WinoRawScratch {
  float* V = nullptr; float* M = nullptr; void* U2 = nullptr; unsigned* words = nullptr;   // words: [B] board ranges, max|U| bits, 1/su
  size_t v_cap = 0, m_cap = 0, u_cap = 0; int b_cap = 0;
  void* w2 = nullptr; unsigned* h2_words = nullptr;   // conv3x3_raw_h2: fp16x2 weight image, [B] board ranges + max|w| bits
  size_t w2_cap = 0; int h2_b_cap = 0, h2_flip = 0;
};
// raw 3x3 convolution with fp16x2 products (direct form; weights split on the device every call)
// ranges != nullptr: [B] bits of max |x| per board, already on the device (the trainer's BatchNorm passes produce them): no range sweep of x here
int conv3x3_raw_h2(agz_ctx* ctx, const float* x, const float* w, float* y, int B, int H, int W, int Cin_p, int Cout_p, WinoRawScratch* sc,
                   const unsigned* ranges = nullptr);
bool conv3x3_raw_h2_fits(int B, int H, int W, int Cin_p, int Cout_p);
// the weight half of conv3x3_raw_h2 alone: max|w| word + the fp16x2 image w2[c/32][tap][piece][n][32] in sc (the trainer's DMA forward
// convolution, train.hip, brings its own activation planes and kernel)
int conv3x3_raw_h2_weights(agz_ctx* ctx, const float* w, int Cin_p, int Cout_p, WinoRawScratch* sc, const void** w2, const unsigned** w_amax);
// The weight halves of the two raw convolutions on a stream of the CALLER's choosing, into buffers the caller owns: the trainer's filters
// do not change inside a step, so it builds every layer's images on its side stream while the forward pass runs (train.hip)
struct RawWeights { void* img = nullptr; unsigned* words = nullptr; size_t cap = 0; };   // words: [0] range bits, [1] Winograd 1 / scale (float), [2..3] spare
int conv3x3_raw_h2_weights_to(agz_ctx* ctx, hipStream_t st, const float* w, int Cin_p, int Cout_p, RawWeights* out);        // words[0] = max|w|
int conv3x3_raw_wino_h2_weights_to(agz_ctx* ctx, hipStream_t st, const float* w, int H, int W, int Cin_p, int Cout_p, RawWeights* out);
void raw_weights_free(RawWeights* rw);
// pre != nullptr: conv3x3_raw_wino_h2_weights_to's image of w (then w is not read)
int conv3x3_raw_wino_h2(agz_ctx* ctx, const float* x, const float* w, float* y, int B, int H, int W, int Cin_p, int Cout_p, WinoRawScratch* sc,
                        const unsigned* ranges = nullptr, const RawWeights* pre = nullptr);
bool conv3x3_raw_wino_h2_fits(int B, int H, int W, int Cin_p, int Cout_p);
void wino_raw_scratch_free(WinoRawScratch* sc);
}  // namespace agz

struct agz_net {
  agz_ctx* ctx = nullptr;
  agz_net_conf conf{};
  std::vector<agz::Param> params;
  std::vector<agz::BNStats> bn;
  bool committed = false;

  // geometry
  int H = 0, W = 0, HW = 0, Hp = 0, Wp = 0;
  int Kp = 0;    // K rounded up to 32
  int Fp = 32;   // input planes padded to 32 channels
  int cfg = 0;   // 0: 128x128 block tile (K%64==0), 1: 128x64 block tile (K%32==0)

  // device parameters (repacked)
  float* d_w_init = nullptr;    // [9][Ntot_init][Fp]
  float* d_w_init_t = nullptr;  // [9][Fp][Ntot_init]: the same filter, channel-contiguous (lat_input_kernel)
  float* d_ep_init = nullptr;   // float2 {scale,shift} [HW][Kp]
  unsigned short* d_w3_init = nullptr;   // bf16x3 image of the input filter [Fp/16][9][3][Kp][16] (cfg 0), conv_x3.hpp
  std::vector<float*> d_w_dual;   // per layer [9][2*Kp][Kp] in block-tile order
  std::vector<float*> d_ep_dual;  // per layer float4 {sa,ta,sb,tb} [HW][Kp]
  std::vector<unsigned short*> d_w3_dual;  // per layer bf16x3 image [Kp/16][9][3][2*Kp][16] (cfg 0 only), conv_x3.hpp
  std::vector<_Float16*> d_w2_dual;        // per layer fp16x2 image [Kp/32][9][2][2*Kp][32] (cfg 0 only), conv_h2.hpp
  std::vector<float> w_unscale;            // per layer 2^-eb of the fp16x2 weight scale
  std::vector<unsigned short*> d_u3_dual;  // per layer Winograd-domain bf16x3 image [36][Kp/16][3][2*Kp][16] (AGZ_COMPUTE_WINO), conv_wino.hpp
  std::vector<_Float16*> d_u2_dual;        // per layer Winograd-domain fp16x2 image [36][Kp/32][2][2*Kp][32] (AGZ_COMPUTE_WINO_H2), conv_wino_h2.hpp
  std::vector<float> u_unscale;            // (unused by the equilibrated image: kept 1)
  std::vector<float*> d_u2_tin;            // per layer [Kp] power-of-two input-channel factors of that image (conv_wino_h2.hpp)
  std::vector<float*> d_u2_colun;          // per layer [2*Kp] 1 / (power-of-two scale of GEMM column n)
  // chained form of that tower (conv_wino_h2c.hpp): the same image in the chained column order / chunk layout, and per block the
  // commit-time bound max|y t_next| <= g1 max|x t_in| + g0 that gives block l+1's operand range before its board maximum exists
  std::vector<_Float16*> d_u2c_dual;
  std::vector<float> wino_g1, wino_g0;
  int wino_gemm = 0;                       // agz_net_set_wino_h2_gemm (agz_debug.h): 0 default, 1 wino_gemm_h2g_kernel, 2 wino_gemm_h2p_kernel
  int wino_form = -1;                      // agz_net_set_wino_h2_form (agz_debug.h): -1 auto (chained where the shape allows), 0 three-kernel block, 1 chained
  void free_u2c() { for (auto& p : d_u2c_dual) if (p) hipFree(p); d_u2c_dual.clear(); wino_g1.clear(); wino_g0.clear(); }
  int build_wino_h2_weights();
  int wino_tm = 4;                         // tile size of that path: 4 = F(4x4,3x3), 5 = F(5x5,3x3), chosen per board size
  size_t wino_v_cap = 0;                   // floats the V scratch holds (fp16x2 path)
  float* d_wV = nullptr;                   // Winograd scratch: transformed input [36][T][Kp] and GEMM output [36][T][2*Kp] of one chunk
  float* d_wM = nullptr;
  int wino_chunk_cap = 0;                  // boards the scratch is sized for
  int build_wino_weights();                // (re)builds d_u3_dual from the host parameters; needs cfg == 0
  unsigned* d_amax = nullptr;              // per-board max |activation|: [B] of the layer about to be consumed (fp16x2), [blocks+1][B] (Winograd fp16x2)
  size_t amax_cap = 0;
  int compute_mode = AGZ_COMPUTE_F32_MFMA;  // agz_net_set_compute_mode
  bool compute_force = false;              // AGZ_COMPUTE_FORCE: split kernels even below the chip-filling threshold
  float* d_head_conv = nullptr;  // [3][Kp] policy ch0, ch1, value ch0 (1x1 filters)
  float* d_head_bn = nullptr;    // [3][HW][2] scale, shift
  float* d_Wp = nullptr;         // [2HW][A]
  float* d_bp = nullptr;         // [A]
  float* d_W1 = nullptr;         // [HW][FC]
  float* d_b1 = nullptr;         // [FC]
  float* d_W2 = nullptr;         // [FC]
  float* d_b2 = nullptr;         // [1]

  // activations (padded NHWC, zero halo)
  int max_batch = 0;
  float* d_act_in = nullptr;  // [B][Hp][Wp][Fp]
  float* d_actA = nullptr;    // [B][Hp][Wp][Kp]
  float* d_actB = nullptr;
  // staging for the host-pointer entry point
  float* d_planes = nullptr;  // [B][F][H][W]
  float* d_policy = nullptr;  // [B][A]
  float* d_value = nullptr;   // [B]
  float* d_ws = nullptr;      // split-K workspace (small batches)
  // AGZ_COMPUTE_WINO_H2 tower: epilogue parameters with the equilibration scales folded in — block l's {scale, shift} pairs multiplied
  // by the NEXT block's t_in (so the activations travel pre-scaled: y' = y t_in, exact powers of two, ReLU commutes) and its scales
  // also by the block's col_unscale; the input convolution's pairs by block 0's t_in.  The transform kernels then load no scale
  // vectors at all.  Host copies of the plain parameters (from commit) are what these are built from.
  std::vector<std::vector<float>> h_ep_dual;
  std::vector<float> h_ep_init;
  std::vector<float*> d_ep_h2;
  float* d_ep_init_h2 = nullptr;
  // latency regime, fp16x2 form (conv_lat.hpp): per layer the equilibrated weight image, t_in[Kp], col_unscale[2Kp]; range words
  std::vector<_Float16*> d_lat_w2;
  std::vector<float*> d_lat_tin, d_lat_colun;
  float* d_lat_wmax[2] = {nullptr, nullptr};   // [B][256] maxima per workgroup, ping-pong between layers
  int lat_wmax_cap = 0;                        // boards
  size_t ws_cap = 0;
  float* d_hs = nullptr;      // latency-regime head scratch: [B][3][HW] features + [B][A+FC] columns
  size_t hs_cap = 0;
  int tower_queues = 0;       // agz_net_set_tower_queues: 0 = auto (two from 256 boards), 1, 2
  bool latency_mode = true;   // allow the small-batch regime (split-K tower + spread heads), agz_net_set_latency_mode

  int ensure_batch(int B);
  void free_device();
  // planes_dev NCHW [B,F,H,W] -> policy_dev [B,A], value_dev [B]; async on ctx stream
  int forward_dev(const float* planes_dev, int B, float* policy_dev, float* value_dev);
  // same but the input already sits in d_act_in (padded NHWC, written by the MCTS encoder)
  int forward_packed(int B, float* policy_dev, float* value_dev);
  // Which kernels a forward of B boards runs (one decision per forward; forward_packed takes exactly these).  Two batch sizes with
  // equal plans run the same kernels, and those are batch-independent bit for bit per board — what lets the engine evaluate a
  // handful of prepareRoot positions as a small batch instead of the arena's whole one (min_same_batch).
  struct FwdPlan { bool half_init, half_dual, small, forced_split, latency, heads_spread, split_ok, heads_nb, wino_wide; };
  FwdPlan fwd_plan(int B) const;
  int min_same_batch(int n, int G) const;   // smallest batch >= n (from n, 16, 32, ... ) with the plan of G; G if there is none
};
