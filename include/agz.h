/*
 * agz.h — C ABI of libagz.so: the MI355X-native replacement for agogo's self-play hot path.
 *
 * Everything here is plain C (opaque handles, plain pointers and sizes, int error codes); no C++
 * or torch types cross this boundary.  Each entry point cites the reference interface it
 * replaces (paths are relative to the gorgonia/agogo tree).  The cgo binding a maintainer would
 * add on the reference side is in INTEGRATION.md / go/agzhip.
 *
 * Threading: an agz_ctx owns one HIP device + one stream and is NOT thread-safe (reference:
 * under `-tags cuda` agogo serialises on a single VM, const_cuda.go:5).  Drive one ctx from one
 * OS thread (Go: runtime.LockOSThread); different ctxs may be driven concurrently.
 *
 * Ownership: the caller allocates every output buffer; the library never returns interior
 * pointers (the reference's Inferencer.Infer returns a slice aliasing the VM output,
 * dualnet/meta.go:186-189 — a latent race this ABI does not reproduce).
 *
 * Errors: 0 = ok, <0 = AGZ_E_*; agz_last_error() returns a thread-local message.
 *
 * Environment switches (ALL of them; each is read once per process, the API default applies when unset, and
 * each has a test that runs it against the default — named in brackets):
 *   AGZ_RCCL_LIB=<path>       library dlopen'ed for the ncclXxx symbols instead of librccl.so (agz_comm_*)
 *                             [tests/test_comm_fake_gpu.py, tests/test_train_gpu.py: the in-tree fake RCCL]
 *   AGZ_WINO_H2_TM=4|5        Winograd tile of AGZ_COMPUTE_WINO_H2: F(4x4,3x3) / F(5x5,3x3); default: the one with
 *                             fewer transform-domain rows for the board   [tests/test_wino_gpu.py knobs test]
 *   AGZ_WINO_H2_CHUNK=<n>     boards per chunk of the AGZ_COMPUTE_WINO_H2 tower (default: as many as 32-bit
 *                             offsets allow); results are bit-identical  [tests/test_wino_gpu.py knobs test]
 *   AGZ_WINO_H2_QUEUES=1|2    overrides agz_net_set_tower_queues         [tests/test_wino_gpu.py knobs test]
 *   AGZ_WINO_H2_FORM=0        the three-kernel block (in / GEMM / out) instead of the chained one (out of block l + in of
 *                             block l+1 in one kernel); same tolerance       [tests/test_wino_gpu.py knobs test]
 *   AGZ_WINO_H2_GEMM=1|2      the GEMM kernel of the chained AGZ_COMPUTE_WINO_H2 block: 1 = 128 x 256 tiles, both operands streamed,
 *                             three workgroups per CU (default); 2 = persistent, the weight slab stationary in registers (K = 256);
 *                             + 64: M / V2c stored with the default cache policy instead of non-temporally (round 4's behaviour);
 *                             all bit-identical                          [tests/test_wino_gpu.py knobs test]
 *   AGZ_WINO_CHUNK=<n>        boards per chunk of the AGZ_COMPUTE_WINO tower; bit-identical
 *                             [tests/test_wino_gpu.py::test_wino_board_chunks_in_a_subprocess]
 */
#ifndef AGZ_H
#define AGZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AGZ_OK 0
#define AGZ_E_INVALID (-1)   /* bad argument / invalid config (reference: Config.IsValid, agogo.go:42-47 panics) */
#define AGZ_E_HIP (-2)       /* HIP runtime failure */
#define AGZ_E_NOMEM (-3)
#define AGZ_E_STATE (-4)     /* call sequence error (e.g. infer before commit) */
#define AGZ_E_TREE_FULL (-5) /* a tree's node pool overflowed (reference cap: MAXTREESIZE, mcts/search.go:23) */
#define AGZ_E_UNSUPPORTED (-6)
#define AGZ_E_CALLBACK (-8)  /* a host inferencer (AGZ_INF_CALLBACK) returned non-zero: the search was aborted */
#define AGZ_E_PEER (-7)      /* a data-parallel step failed on ANOTHER rank (agz_trainer_forward_backward_allreduce): gradients undefined */

typedef struct agz_ctx agz_ctx;
typedef struct agz_net agz_net;
typedef struct agz_arena agz_arena;

/* game.Colour / game.Player values (game/state.go:9-13) */
#define AGZ_NONE 0
#define AGZ_BLACK 1
#define AGZ_WHITE 2
/* game.Single special moves (mcts/mcts.go:21-22) */
#define AGZ_PASS (-1)
#define AGZ_RESIGN (-2)

const char* agz_last_error(void);
/* library / build info string ("libagz <ver> gfx950 ...") */
const char* agz_version(void);

/* ---- context ------------------------------------------------------------------------------ */
/* One HIP device + stream.  Replaces the implicit gorgonia VM/engine an Agent owns (agent.go:44-53). */
int agz_ctx_create(int device, agz_ctx** out);
void agz_ctx_destroy(agz_ctx* ctx);
/* Page-locked host memory for the buffers handed to the host-pointer entry points (agz_net_infer, agz_arena_get_*, ...):
 * the library stages pageable memory through the driver's bounce buffer; a buffer from agz_host_alloc is DMA-ed directly
 * (the batch-1 agogo.Inferer path — agent.go:60-74 hands one board per call — is pure latency otherwise).  Plain host
 * memory as far as the caller is concerned; free with agz_host_free before the ctx goes. */
int agz_host_alloc(agz_ctx* ctx, size_t bytes, void** out);
int agz_host_free(agz_ctx* ctx, void* p);
int agz_ctx_sync(agz_ctx* ctx);
/* raw hipStream_t the ctx launches on (for callers that want to record their own events) */
void* agz_ctx_stream(agz_ctx* ctx);
/* (kernel-class timers and test diagnostics: include/agz_debug.h — not part of the drop-in surface) */
/* ---- dual network ------------------------------------------------------------------------- */
/* Fields 1:1 with dual.Config (dualnet/config.go:4-16); bn_* pin the BatchNorm inference
 * semantics that live in un-vendored gorgonia. */
#define AGZ_BN_DEGENERATE_EPS 0 /* x / sqrt(0 + eps) * gamma + beta  (op.Reset() before every Infer, meta.go:170-172) */
#define AGZ_BN_RUNNING 1        /* (x - mean_c) / sqrt(var_c + eps) * gamma + beta, stats via agz_net_set_bn_stats */
#define AGZ_BN_IDENTITY 2       /* x * gamma + beta */
typedef struct agz_net_conf {
  int32_t K;            /* filters */
  int32_t SharedLayers; /* dual-branch blocks */
  int32_t FC;           /* value hidden width */
  int32_t BatchSize;    /* only used for Glorot fan computation of batch-shaped params (App. B b5) */
  int32_t Width, Height;
  int32_t Features;
  int32_t ActionSpace;  /* policy width (moves + pass) */
  int32_t bn_mode;      /* AGZ_BN_* */
  float bn_eps;         /* 1e-5 in the reference (ermahagerdmonards.go:54) */
} agz_net_conf;

/* dual.New + (*Dual).Init (dualnet/dual.go:33-47). Parameters are created zeroed. */
int agz_net_create(agz_ctx* ctx, const agz_net_conf* conf, agz_net** out);
void agz_net_destroy(agz_net* net);
/* (*Dual).Model() (dualnet/dual.go:134-142): learnables in creation order of Dual.fwd:
 *   Init{filter,gamma,beta}, per shared layer {L1 filter,gamma,beta, L2 filter,gamma,beta},
 *   PolicyHead{filter,gamma,beta}, Policy_w, Policy_b, ValueHead{filter,gamma,beta}, Value_w, Value_b,
 *   ValueOutput_w, ValueOutput_b.
 * Row-0 shapes are stored: conv filter [out,in,k,k]; BN gamma/beta [C,H,W]; FC w [in,units]; FC b [units]. */
int agz_net_num_params(const agz_net* net);
int agz_net_param_info(const agz_net* net, int index, char* name, size_t name_cap, size_t* n_elems);
/* G.Let on a learnable / the copy loop of dual.Infer (dualnet/meta.go:141-146).  n may exceed the
 * row-0 size (batch-shaped reference tensors): the first n_elems floats (= row 0) are taken, as Go
 * copy() + "only row 0 is ever real" does (App. B b5).  For BN params n == C broadcasts per channel. */
int agz_net_set_param(agz_net* net, int index, const float* host, size_t n);
int agz_net_get_param(const agz_net* net, int index, float* host, size_t n);
/* per-channel running statistics for AGZ_BN_RUNNING; bn_index counts BN ops in Dual.ops order
 * (dual.go:59-100: Init, L1/L2 of each shared layer, then policy, value). */
int agz_net_set_bn_stats(agz_net* net, int bn_index, const float* mean, const float* var, size_t C);
/* Random init with the reference's initialiser KINDS (GlorotU conv, GlorotN FC + BN gamma/beta,
 * zero FC bias; ermahagerdmonards.go:39,80,82) from the build's own RNG (Go's math/rand stream is
 * not reproducible here). */
int agz_net_init_random(agz_net* net, uint64_t seed);
/* Upload + repack to device layouts (folds BN into per-(c,h,w) scale/shift). Must follow any set_param. */
int agz_net_commit(agz_net* net);
/* dual.Infer + (*Inferencer).Infer (dualnet/meta.go:125-190), batched: planes [B,F,H,W] fp32 NCHW
 * (every board evaluated with row-0 parameters), policy [B,ActionSpace] softmax, value [B] tanh.
 * Host buffers (pageable ok). */
int agz_net_infer(agz_net* net, const float* planes, int B, float* policy, float* value);
/* same with device pointers, asynchronous on the ctx stream */
int agz_net_infer_dev(agz_net* net, const float* planes_dev, int B, float* policy_dev, float* value_dev);
/* Small-batch ("latency") regime for tournament-style single-tree Agent.Search (agent.go:76-81: one board per
 * Infer call): when a forward's tower would occupy <= 1/4 of the CUs, the convolutions run split-K and the heads are
 * spread over the chip.  On by default.  Within a regime results are bit-identical for every batch size; ACROSS the
 * two regimes the fp32 summation order differs (same tolerance vs the reference).  In AGZ_COMPUTE_F32_MFMA (the default mode)
 * the small-batch tower stays on exact fp32 products (split-K); in every other mode (AGZ_COMPUTE_AUTO included) K = 64 / 128 /
 * 256 towers take the one-launch-per-layer kernel with fp16x2 products on equilibrated operands (hi/lo fp16 pieces, the lo*lo
 * term of 2^-22 relative dropped; conv_lat.hpp) — same tolerance, 2.6x faster per layer at batch 1.  Turn the regime off for
 * strict bitwise batch-size independence at every batch size. */
int agz_net_set_latency_mode(agz_net* net, int on);
/* Queues the Winograd fp16x2 tower (AGZ_COMPUTE_WINO_H2) runs on: 2 = the batch is split into two halves whose block chains
 * run on two HIP streams, so one half's HBM-bound transform kernels run under the other half's GEMM (boards are independent:
 * results are bit-identical to one queue); 1 = one launch chain over the whole batch; 0 = auto (two queues from 256 boards
 * on, the default).  Per-kernel durations inflate under the overlap: measure kernels with 1. */
int agz_net_set_tower_queues(agz_net* net, int queues);
/* Arithmetic of the dual-block convolutions (both are fp32-grade and meet the same parity tolerance):
 *   AGZ_COMPUTE_F32_MFMA  v_mfma_f32_32x32x2_f32, exact fp32 products (default)
 *   AGZ_COMPUTE_BF16X3    every fp32 operand split exactly into three bf16 pieces, six bf16 MFMAs per product
 *                         (dropped cross terms <= 2^-23 relative), fp32 accumulation — 2.67x the fp32 matrix rate.
 *   AGZ_COMPUTE_FP16X2    every operand scaled by a power of two into fp16 range and split into two fp16 pieces
 *                         (23 significand bits), three fp16 MFMAs per product, fp32 accumulation; the scale of
 *                         every board's activations is measured on the device per layer (results stay independent
 *                         of the batch composition).  Same parity tolerance on the tested nets; elements more than
 *                         2^17 below their board's maximum lose relative precision — opt-in.
 *                         Both split modes apply to K a multiple of 64 and batches whose 128-row tiles fill the chip
 *                         (>= one tile per CU); other shapes keep F32_MFMA, which is faster there. */
#define AGZ_COMPUTE_F32_MFMA 0
#define AGZ_COMPUTE_BF16X3 1
#define AGZ_COMPUTE_FP16X2 2
#define AGZ_COMPUTE_WINO 3 /* Winograd F(4x4,3x3): transforms in fp32, the 36 transform-domain GEMMs with BF16X3 products — 3.6x
                            * fewer matrix instructions on 19x19; rounding error ~5x a direct fp32 convolution's (still inside
                            * the stated tolerance); opt-in, same shape conditions as the split modes */
#define AGZ_COMPUTE_AUTO 4 /* the measured choice: WINO_H2 where its weight image exists (K a multiple of 64), else BF16X3 where the
                            * split kernels apply, else F32_MFMA   [tests/test_wino_gpu.py::test_compute_auto_takes_the_measured_mode] */
#define AGZ_COMPUTE_WINO_H2 5 /* Winograd F(5x5,3x3) / F(4x4,3x3) (whichever needs fewer rows for the board) with FP16X2 products in the transform domain: the input transform writes the
                            * operand already split into two fp16 pieces (scaled per board by a power of two from a proven bound,
                            * overflow impossible), three fp16 MFMAs per product — half the matrix instructions of AGZ_COMPUTE_WINO */
#define AGZ_COMPUTE_FORCE 0x100 /* OR-ed in: take the split kernel even below the chip-filling threshold (tests) */
int agz_net_set_compute_mode(agz_net* net, int mode);
/* Checkpoint of the learnables in Model() order (+ BN statistics).  The reference gob-encodes G.Values
 * (AZ.Save / Dual.GobEncode, agogo.go:175-209, dualnet/dual.go:180-206); gob is Go-only, so this is a documented
 * flat format: "AGZNET01", agz_net_conf, n_params, then per parameter {uint64 n, float32[n]}, then per BN op
 * {uint64 C, mean[C], var[C]}.  agz_net_load requires an identical conf and leaves the net committed. */
int agz_net_save(const agz_net* net, const char* path);
int agz_net_load(agz_net* net, const char* path);
/* FLOPs of one evaluation (SURVEY App. D formula) */
double agz_net_flops_per_eval(const agz_net* net);

/* ---- dual.Train ---------------------------------------------------------------------- */
typedef struct agz_trainer agz_trainer;
/* The training graph of Dual.fwd + Dual.bwd (dualnet/dual.go:50-132) for conf->BatchSize rows.  Learnables keep the
 * reference's FULL shapes in Model() order: conv filter [out,in,k,k]; BN gamma/beta [B,C,H,W]; FC w [in,units];
 * FC b [B,units]  (batch-shaped).  BatchNorm runs in training mode (batch statistics). */
int agz_trainer_create(agz_ctx* ctx, const agz_net_conf* conf, agz_trainer** out);
void agz_trainer_destroy(agz_trainer* t);
int agz_trainer_num_params(const agz_trainer* t);
int agz_trainer_param_info(const agz_trainer* t, int index, char* name, size_t name_cap, size_t* n_elems);
int agz_trainer_set_param(agz_trainer* t, int index, const float* host, size_t n);
int agz_trainer_get_param(const agz_trainer* t, int index, float* host, size_t n);
int agz_trainer_get_grad(const agz_trainer* t, int index, float* host, size_t n);
int agz_trainer_init_random(agz_trainer* t, uint64_t seed);
/* One batch of dual.Train's inner loop (dualnet/meta.go:33-40): Let planes/Pi/V, RunAll, solver.Step(lr).
 * planes [B,F,H,W], pi [B,ActionSpace], v [B] host buffers; *cost = xent(logits,Pi) + mean((o-V)^2) (dual.go:113-121).
 * NOTE (gradient read-back): with lr != 0 this call — and agz_train / agz_train_dev, which are loops of it — takes the SGD step of the
 * tower's batch-shaped gamma / beta INSIDE the BatchNorm backward kernel (98 % of the learnables: no gradient round trip), so their
 * gradients are NOT materialised: agz_trainer_get_grad / agz_trainer_grads_dev then return what an earlier forward_backward left there
 * for those tensors (zeros on a fresh trainer); filter and head gradients are current.  To read every gradient, run
 * agz_trainer_forward_backward (or agz_trainer_batch with lr = 0).  A step that fails part-way may have stepped some layers' gamma / beta
 * already: treat the trainer's parameters as undefined after an error (reload a checkpoint, agz_trainer_load). */
int agz_trainer_batch(agz_trainer* t, const float* planes, const float* pi, const float* v, float lr, float* cost);
/* Split form for data-parallel training: forward_backward fills the flat gradient buffer; all-reduce it over RCCL
 * (agz_trainer_grads_dev gives the device pointer: ONE collective per step); apply does w -= lr*grad_scale*grad. */
int agz_trainer_forward_backward(agz_trainer* t, const float* planes, const float* pi, const float* v, float* cost);
/* the same on device buffers (e.g. one batch of agz_examples_tensors_dev); cost may be NULL (no synchronisation) */
int agz_trainer_forward_backward_dev(agz_trainer* t, const float* planes_dev, const float* pi_dev, const float* v_dev, float* cost);
int agz_trainer_apply(agz_trainer* t, float lr, float grad_scale);
int agz_trainer_grads_dev(agz_trainer* t, float** dev_ptr, size_t* n_floats);
/* Arithmetic of training's three GEMMs (forward convolution, data gradient, weight gradient): AGZ_COMPUTE_F32_MFMA (default),
 * AGZ_COMPUTE_BF16X3 (all three on the bf16 pipe; weights re-split on the device every step) or AGZ_COMPUTE_WINO_H2 (fp16x2
 * products throughout: forward = the DIRECT 3x3 convolution with fp16 hi/lo operands, weight image split on the device every step;
 * data gradient = the Winograd fp16x2 path with device-transformed weights; weight gradient = fp16 hi/lo operands split once per
 * layer; the first layer, 18 -> K, stays on BF16X3).  130 / 88 / 50 ms per step at 19x19, K = 256, 20 blocks, batch 256.  Every mode
 * meets the same gradient tolerance against the reference arithmetic — every gradient tensor within 2e-5 of its maximum, tested per
 * mode at the headline width (K = 256, 19x19) on several data draws — which is why the forward convolutions do NOT take the
 * Winograd path: its rounding (2e-6 of the output rms against 3e-7 for the direct forms) puts a pre-activation on the other side of
 * a ReLU than the reference on every other batch at that width. */
int agz_trainer_set_compute_mode(agz_trainer* t, int mode);
/* dual.Train(d, Xs, policies, values, batches, iterations) (dualnet/meta.go:16-54): lr 0.1 vanilla SGD, shuffleBatch
 * after every iteration (build RNG; Xs/policies/values are shuffled in place like the reference). */
int agz_train(agz_trainer* t, float* Xs, float* policies, float* values, int batches, int iterations, uint64_t seed,
              float* last_cost);
/* dual.Train over DEVICE tensors (e.g. agz_examples_tensors_dev): same loop and shuffleBatch stream as agz_train; the
 * shuffle permutes 4-byte row indices, batches are gathered device to device, the cost is read back once at the end.
 * The tensors themselves are left in place (the reference shuffles rows in place, meta.go:57-102; AZ.Learn discards
 * them right after, agogo.go:133). */
int agz_train_dev(agz_trainer* t, const float* Xs_dev, const float* policies_dev, const float* values_dev, int batches,
                  int iterations, uint64_t seed, float* last_cost);
/* Checkpoint of the trainable network in its full batch-shaped form (AZ.Save/AZ.Load agogo.go:175-209 for the side that
 * keeps learning; gob is Go-only, the format is documented in train.hip). load: the file must match the configuration. */
int agz_trainer_save(const agz_trainer* t, const char* path);
int agz_trainer_load(agz_trainer* t, const char* path);
/* dual.Infer's copy loop (dualnet/meta.go:141-146): row 0 of every learnable -> the inference net; commits it. */
int agz_trainer_export(const agz_trainer* t, agz_net* net);

/* ---- batched self-play arenas: game.State + mcts.MCTS + agogo.Arena on device ---------------- */
#define AGZ_GAME_MNK 0  /* game/mnk  (m,n,k) */
#define AGZ_GAME_C4 1   /* game/c4   (rows=m, cols=n, k in a row) */
#define AGZ_GAME_KOMI 2 /* game/komi (m,n, capture-k) */
#define AGZ_GAME_WQ 3   /* game/wq   (m=n=size, komi) */
typedef struct agz_game_conf {
  int32_t kind;
  int32_t m, n, k;
  float komi;        /* wq: AdditionalScore (game/wq/game.go:181) */
  int32_t max_moves; /* arena safety cap on game length (the reference has no ko rule; 0 = 2*m*n) */
  int32_t encoder;   /* AGZ_ENC_* */
} agz_game_conf;
#define AGZ_ENC_TWOPLANE 0 /* cmd/tictactoe/main.go:26-47 : board(+-1, 0->0.001) + to-move plane, F=2 */
#define AGZ_ENC_WQ 1       /* WQEncoder, encoding_helper.go:29-68, F=18 */

/* mcts.Config (mcts/tree.go:15-29); Timeout is replaced by "exactly Budget simulations"  */
#define AGZ_DONT_PREFER_PASS 0
#define AGZ_PREFER_PASS 1
#define AGZ_DONT_RESIGN 2
typedef struct agz_mcts_conf {
  float PUCT;
  int32_t M, N;
  int32_t RandomCount;
  int32_t Budget;
  uint32_t RandomMinVisits;
  float RandomTemperature;
  int32_t DumbPass;
  float ResignPercentage;
  int32_t PassPreference;
} agz_mcts_conf;

/* Inferencer kinds an Agent can hold (mcts.Inferencer, mcts/mcts.go:15-18) */
#define AGZ_INF_NET 0    /* Agent.Infer over a dual net (agent.go:60-74) */
#define AGZ_INF_DUMMY 1  /* agogo.dummyInferer: uniform 1/ActionSpace policy, value +1 Black / -1 White agent (dummy.go:10-23) */
#define AGZ_INF_SCRIPT 2 /* mcts/example_test.go:40-72 dummyNN (tic-tac-toe script by move number) */
#define AGZ_INF_HASH 3   /* deterministic synthetic inferencer (integer hash of the position): parity tests */
#define AGZ_INF_UNIFORM 4 /* mcts/example_test.go:158-166 dummyNN2: 1/25 policy (len 25), value 1/25 */
#define AGZ_INF_CALLBACK 5 /* ANY mcts.Inferencer as a host function (agz_arena_set_inferencer_callback / agz_mcts_set_inferencer_callback) */

/* The `nn Inferencer` argument of mcts.New is an interface: Infer(state game.State) (policy []float32, value float32)
 * (mcts/mcts.go:15-18; tree.go:80).  A host inferencer meets the device search between its two kernels: after the descent every
 * expandable leaf of the agents that hold one is handed over as ONE batch — per leaf the encoder's input tensor (what Agent.Infer builds from
 * the state, agent.go:60-74), the board, the mover and MoveNumber() — and the callee fills one policy row and one value per leaf, exactly
 * what Infer returns: policy_len probabilities whose LAST entry is the pass probability (search.go:276), and the value as the
 * network reports it (the search itself turns it into `1 - value` for White, search.go:278-280).  The rows are consumed by the same
 * expansion kernel that reads a network's.  One host round trip per simulation step for all games of the arena: the boundary for
 * caller-supplied networks and a network-independent differential hook; AGZ_INF_NET keeps the whole step on the device.
 * Return 0; anything else aborts the search (AGZ_E_CALLBACK; reset the arena).  The function runs on the thread that called into libagz. */
typedef struct agz_leaf_batch {
  int32_t n;                  /* leaves in this call (>= 1) */
  int32_t features, height, width; /* geometry of `planes` (the arena's encoder) */
  int32_t policy_len;         /* floats per policy row (the value given at registration) */
  const float* planes;        /* [n][features][height][width] */
  const int32_t* board;       /* [n][height*width]  AGZ_NONE / AGZ_BLACK / AGZ_WHITE */
  const int32_t* to_move;     /* [n] the leaf state's ToMove() */
  const int32_t* move_number; /* [n] the leaf state's MoveNumber() */
  const int32_t* game;        /* [n] which game of the arena the leaf belongs to */
  float* policy;              /* OUT [n][policy_len] */
  float* value;               /* OUT [n] */
} agz_leaf_batch;
typedef int (*agz_infer_fn)(void* user, const agz_leaf_batch* batch);

/* MakeArena × n_games (arena.go:42-70): n_games independent games, each with agents A and B, each
 * agent with its own search tree (mcts.New, mcts/tree.go:80-103).  max_nodes = node-pool capacity per
 * tree; 0 = the default: FOUR searches' worth of expansions, 4 * (Budget + 2) * (ActionSpace + 1) nodes (at most 8 M) — a search adds at most
 * Budget + 1 expansions to the subtree kept from the one before, so a tree that keeps a fraction f of its nodes per move settles at
 * (Budget + 1)(ActionSpace + 1) / (1 - f): the default covers f <= 0.75.  A very NARROW tree keeps more and can outgrow it (the reference's
 * arena is unbounded up to MAXTREESIZE, search.go:23,78).  With max_nodes = 0 — the library chose the size — the pools therefore GROW when a
 * search could outgrow them (AGZ_POOL_GROW below is the default policy then: results unchanged, overflow impossible short of AGZ_E_NOMEM).  An
 * explicit max_nodes > 0 is the caller's memory budget (AGZ_POOL_STRICT): a tree whose pool fills stops growing for that move,
 * agz_arena_stats.tree_full counts it and agz_arena_play / agz_arena_selfplay return AGZ_E_TREE_FULL.  Examples: up to 2 * n_games * max_moves rows, at most 12 GiB per arena
 * (agz_arena_stats.examples_dropped counts rows that did not fit).  Budget 0 with several games: no simulations at all, every move
 * comes from prepareRoot (search.go:392-408). */
int agz_arena_create(agz_ctx* ctx, const agz_game_conf* game, const agz_mcts_conf* mcts, int n_games,
                     uint64_t seed, int max_nodes, agz_arena** out);
void agz_arena_destroy(agz_arena* arena);
/* Agent.NN / SwitchToInference / useDummy (agent.go:42-57,105-113): agent 0 = A, 1 = B. net may be NULL
 * for the non-NET kinds. */
int agz_arena_set_inferencer(agz_arena* arena, int agent, int kind, agz_net* net);
/* agent `agent` holds a host inferencer (AGZ_INF_CALLBACK, above).  policy_len >= the game's ActionSpace (<= 4096).  Setting another
 * kind with agz_arena_set_inferencer removes it. */
int agz_arena_set_inferencer_callback(agz_arena* arena, int agent, agz_infer_fn fn, void* user, int policy_len);
/* What a FULL node pool means (default: AGZ_POOL_GROW when the arena was created with max_nodes = 0, AGZ_POOL_STRICT with an explicit
 * max_nodes).  AGZ_POOL_STRICT: the tree stops growing, agz_arena_stats.tree_full counts it, and agz_arena_play /
 * agz_arena_selfplay / agz_mcts_search fail with AGZ_E_TREE_FULL — nothing is silently truncated, and every parity test runs this way.
 * AGZ_POOL_STOP_SEARCH: the reference's own rule — a tree at MAXTREESIZE stops being searched for that move and the game goes on
 * (search.go:23,78,229) — with max_nodes in MAXTREESIZE's place: the move is the best of the truncated search, the next move re-roots into
 * the free pool, tree_full still counts.  For long unattended self-play with a peaked network, where one narrow tree should not end the run. */
/* AGZ_POOL_GROW: pools that cannot overflow.  Before every search (agz_arena_begin_move; again inside agz_arena_simulate when a move runs more
 * simulations than its Budget) the host reads every tree's node count and, when the fullest tree could outgrow its pool — a search adds at
 * most (simulations + 1) x (ActionSpace + 1) nodes — re-allocates all pools at a larger capacity and copies the live trees over: node indices
 * and every result stay what they were (bit-exact against AGZ_POOL_STRICT with a large enough max_nodes).  One small read-back per move; a
 * re-allocation takes tens of milliseconds and is rare.  AGZ_E_NOMEM when the device cannot hold the larger pools.  A wall-clock search
 * (Budget <= 0) is covered slice by slice. */
#define AGZ_POOL_STRICT 0
#define AGZ_POOL_STOP_SEARCH 1
#define AGZ_POOL_GROW 2
int agz_arena_set_pool_policy(agz_arena* arena, int policy);
/* Start new games: fresh trees, empty boards, colour assignment.  a_is_black: per game 0/1, or NULL to
 * draw it from the arena RNG (arena.go:81-89 draws a.r.Intn(2)). */
int agz_arena_reset(agz_arena* arena, const uint8_t* a_is_black);
/* Arena.Play loop body for every unfinished game (arena.go:96-138): current agent Search
 * (mcts/search.go:92-164) -> record example -> Apply -> switch player -> Ended / two passes.
 * n_moves = how many plies to advance (<=0: until all games ended). */
int agz_arena_play(agz_arena* arena, int n_moves, int record);
/* Continuous self-play: like agz_arena_play, but a finished game is replaced at once by a fresh one (new trees,
 * colours drawn again) so all n_games slots stay busy; returns once n_games_target games have finished since the
 * last reset.  This is the `for e < episodes { ex = append(ex, a.SelfPlay()...) }` loop of AZ.Learn
 * (agogo.go:110-114) with the episodes running concurrently.  Needs both agents on one net (or synthetic). */
int agz_arena_selfplay(agz_arena* arena, int64_t n_games_target, int record);
/* Finer steps of one ply, for benchmarks and parity tests:
 *   begin_move   = updateRoot + prepareRoot (search.go:94-109)
 *   simulate(k)  = k x { pipeline (search.go:209-257) for every game, leaves coalesced into one
 *                  batched inference }
 *   end_move     = bestMove + Policies + Apply (search.go:151-161, arena.go:105-138) */
/* BUILD EXTENSION (no mcts.Config field): the simulations of one tree run in rounds of `lanes` (1..16) whose leaves are
 * evaluated as one batch — the deterministic, lane-ordered form of the reference's NumCPU goroutines sharing a tree with
 * the stored virtual loss (search.go:112-131, node.go:147-159,248-260; restated in oracle/mcts.hpp parallelRound, which
 * the device matches bit for bit).  For tournament latency: a single tree no longer evaluates one board at a time.
 * 1 (default) is the sequential search, the declared semantics of everything else in this library.  Changes the search
 * result (different, not worse, simulations); call between searches. */
int agz_arena_set_parallel(agz_arena* arena, int lanes);
int agz_arena_begin_move(agz_arena* arena);
int agz_arena_simulate(agz_arena* arena, int k);
int agz_arena_end_move(agz_arena* arena, int record);
/* Apply externally chosen moves instead of searching — moves[g] (a game.Single: cell / column, AGZ_PASS, AGZ_RESIGN) for
 * every game, ignored for finished games.  This is the opponent's reply in a tournament: the caller of Agent.Search
 * (agent.go:76-81) applies the opponent's move to its game.State and hands the new state to the next Search; here the
 * state lives on the device.  No example is recorded; the next search of either agent re-roots its tree by replaying
 * the moves played since its last search (updateRoot, search.go:424-500).  Every move must pass State.Check: a game
 * with an illegal move is left unchanged and AGZ_E_INVALID is returned (the other games are applied); AGZ_NO_MOVE
 * skips a game (e.g. to re-send a corrected move for one game only).  With two different nets the arena evaluates the
 * games in lockstep plies: keep all unfinished games at the same ply. */
#define AGZ_NO_MOVE (-32768)
int agz_arena_apply_moves(agz_arena* arena, const int32_t* moves);
/* BUILD EXTENSION — synthetic openings for benchmarks and parity tests ("for each slot play u uniformly-random
 * legal moves from the empty board"): game g plays n_moves[g] uniformly drawn legal board moves for alternating colours (Pass
 * only when the mover has no legal board move and the game has a pass), like agz_arena_apply_moves: no search, no example,
 * trees re-root at their next search.  Deterministic in (seed, g, moves played so far); restated by the oracle. */
int agz_arena_random_moves(agz_arena* arena, const int32_t* n_moves, uint64_t seed);

/* A game.State as the host holds it (game/state.go:125-156) — what mcts.SetGame / Agent.Search receive (tree.go:120-124,
 * agent.go:77-80) when the position was NOT reached by playing on the device. */
typedef struct agz_state {
  const int32_t* board;       /* [m*n] game.Colour per cell (State.Board()) */
  int32_t to_move;            /* State.ToMove() */
  int32_t n_moves;            /* moves applied so far = len(history) (State.MoveNumber(); c4 reports 1 regardless) */
  int32_t passes;             /* State.Passes(): consecutive passes so far (wq) */
  uint32_t hash;              /* State.Hash() for komi/wq (the running zobrist hash; mnk/c4 hash the board itself) */
  float captures_black, captures_white; /* komi: stones captured by each side (State.Score) */
  const int32_t* last_moves;  /* the most recent n_last_moves moves, oldest first (LastMove()/UndoLastMove() chain): tree reuse */
  int32_t n_last_moves;       /* replays them (search.go:424-469); with fewer than the plies since the previous Search a fresh root is built */
  const int32_t* historical;  /* [n_historical][m*n] boards after the last n_historical moves, oldest first (State.Historical): */
  int32_t n_historical;       /* WQEncoder's history planes (encoding_helper.go:29-68), at most 8 */
} agz_state;
/* overwrite game g of an arena with a host-side state (trees are kept: the next search re-roots or starts fresh) */
int agz_arena_set_state(agz_arena* arena, int g, const agz_state* st);

/* --- observers (all copy into caller buffers) --- */
typedef struct agz_arena_stats {
  int64_t sims_total;    /* pipeline invocations (iter, search.go:181) */
  int64_t sims_nonnull;  /* playouts: non-null results (search.go:175-178) */
  int64_t nn_evals;      /* inferencer evaluations */
  int64_t moves_played;
  int64_t games_finished;
  int64_t examples;
  int32_t n_games;
  int32_t n_active;      /* games not yet ended */
  int32_t tree_full;     /* number of trees that overflowed their pool */
  int32_t examples_dropped; /* examples lost because the arena's example buffer was full (clear or append them earlier) */
  int64_t path_nodes;    /* measurement: nodes on the selected paths, summed over simulations (mean depth = / sims_total) */
  int64_t children_read; /* measurement: children Node.Select read, summed over simulations (12 B each, node.go:170-237) */
} agz_arena_stats;
int agz_arena_get_stats(agz_arena* arena, agz_arena_stats* out);
/* Agent statistics since the last reset (Agent.Wins / Loss / Draw, arena.go:156-171; Statistics, statistics.go):
 * A's wins = B's losses and vice versa.  AZ.Learn's gating rule is b_wins / (b_wins + a_wins) > UpdateThreshold
 * (agogo.go:155). */
int agz_arena_get_results(agz_arena* arena, int64_t* a_wins, int64_t* b_wins, int64_t* draws);
/* game state of game g: board [m*n] colours, to_move, move number, passes, ended, winner, a_is_black,
 * last best move */
typedef struct agz_game_state {
  int32_t to_move, move_number, passes, ended, winner, a_is_black, last_move, reserved;
  float score_black, score_white;
} agz_game_state;
int agz_arena_get_game(agz_arena* arena, int g, int32_t* board, agz_game_state* st);
/* moves played so far in game g (game.Single per ply); returns count via *n */
int agz_arena_get_history(agz_arena* arena, int g, int32_t* moves, int cap, int* n);
/* root children of agent's tree in game g after a search, in bestMove's fancySort order
 * (search.go:353): move, visits, blackScores, prior. Returns the number of children in *n. */
int agz_arena_root_children(agz_arena* arena, int g, int agent, int32_t* moves, uint32_t* visits,
                            float* black_scores, float* priors, int cap, int* n);
/* mcts.Nodes() analogue: nodes allocated in the tree pool */
int agz_arena_tree_nodes(agz_arena* arena, int g, int agent, int* n_nodes);
/* examples recorded so far (arena.go:105-123,146-155): planes [n, F*m*n], policy [n, A+1], value [n]
 * (labelled +1/-1/0 once the game has ended; before that the raw mover colour 1/2), game index [n].
 * Pass NULL buffers to query the count. */
int agz_arena_get_examples(agz_arena* arena, float* planes, float* policy, float* value, int32_t* game_idx,
                           int cap, int* n);
int agz_arena_clear_examples(agz_arena* arena);
/* Removes the examples of FINISHED games (labelled rows) from the arena's buffer and keeps the rows of games still in flight
 * (compacted, their per-game chains re-linked), so that continuous self-play can be harvested repeatedly without duplicates
 * and without losing the earlier plies of running games.  agz_examples_append_arena calls it (take semantics);
 * agz_arena_clear_examples, in contrast, discards everything — including the rows of running games. */
int agz_arena_drop_labelled_examples(agz_arena* arena);
/* device pointers of the example buffers (for an RCCL all-gather before dual.Train) */
int agz_arena_examples_dev(agz_arena* arena, float** planes, float** policy, float** value, int* n);
/* device flags [n]: 1 = the example's game has ended and Value is the +1/-1/0 label; 0 = still the raw mover colour */
int agz_arena_examples_labelled_dev(agz_arena* arena, const uint8_t** labelled);

/* ---- mcts.MCTS: ONE search tree on a caller-owned game.State (mcts/tree.go:80-142, mcts/search.go:92-164) ----------------
 * The drop-in for `mcts.New(game, conf, nn)` behind Agent.Search (agent.go:77-80): the host keeps its game.State, hands the
 * position over with agz_mcts_set_game and applies the returned move itself.  The tree persists between searches and is
 * re-rooted (updateRoot, search.go:424-500) when the new position follows from the previous one by agz_state.last_moves. */
typedef struct agz_mcts agz_mcts;
/* mcts.New (tree.go:80-103).  max_nodes: node-pool capacity (0 = default from Budget and the action space). */
int agz_mcts_create(agz_ctx* ctx, const agz_game_conf* game, const agz_mcts_conf* conf, uint64_t seed, int max_nodes, agz_mcts** out);
void agz_mcts_destroy(agz_mcts* mcts);
/* the `nn Inferencer` argument of mcts.New (mcts/mcts.go:15-18): AGZ_INF_* (net may be NULL for the synthetic kinds) */
int agz_mcts_set_inferencer(agz_mcts* mcts, int kind, agz_net* net);
/* mcts.New(game, conf, nn) with a caller-supplied Inferencer: `fn` is called once per simulation with the one leaf state (lane rounds:
 * up to `lanes` leaves) — the reference's own dummyNN (mcts/example_test.go:40-72) runs through this as it runs through mcts.New */
int agz_mcts_set_inferencer_callback(agz_mcts* mcts, agz_infer_fn fn, void* user, int policy_len);
/* AGZ_POOL_* for a single tree (see agz_arena_set_pool_policy) */
int agz_mcts_set_pool_policy(agz_mcts* mcts, int policy);
/* lanes per round (BUILD EXTENSION, see agz_arena_set_parallel) */
int agz_mcts_set_parallel(agz_mcts* mcts, int lanes);
/* (*MCTS).SetGame (tree.go:120-124) */
int agz_mcts_set_game(agz_mcts* mcts, const agz_state* st);
/* (*MCTS).Search(player) (search.go:92-164): SetToMove(player), updateRoot, prepareRoot, Budget simulations, bestMove,
 * prev = current.Clone(), cachedPolicies[{hash, best}]++.  The game is NOT advanced.  *best: cell / column, AGZ_PASS, AGZ_RESIGN. */
int agz_mcts_search(agz_mcts* mcts, int player, int32_t* best);
/* (*MCTS).Policies(current game) (tree.go:128-142): [ActionSpace+1] floats; NaN when no search of this position is cached */
int agz_mcts_policies(agz_mcts* mcts, float* policy, int cap);
/* root children after a search in bestMove's order (the debugging surface of (*MCTS).Children, unsafe_safe.go:15) */
int agz_mcts_root_children(agz_mcts* mcts, int32_t* moves, uint32_t* visits, float* black_scores, float* priors, int cap, int* n);
/* (*MCTS).Children(of) with the Node fields (*MCTS).Log / ToDot print (mcts/unsafe_safe.go:15, graph.go:34): the children of ANY
 * node of the live tree — node 0 is the root, child_ids feed further calls; *n = number of children (0: not expanded) */
int agz_mcts_children(agz_mcts* mcts, int node, int32_t* child_ids, int32_t* moves, uint32_t* visits, float* black_scores,
                      float* priors, int cap, int* n);
/* (*MCTS).ToDot() (mcts/graph.go:34-90): the live tree as Graphviz text — digraph "G", one HTML-table node per tree node with the
 * reference's rows (Node ID, Move, Player, Visits, Score = prior, State = the moves of its path on an empty board; "Value", the
 * evaluation a node was created with, is not kept on the device and prints as "-"), children in move order.  max_nodes > 0 limits
 * the output to the first max_nodes nodes of the pool (a top of the tree); *needed = bytes incl. the terminating 0 — call with
 * cap = 0 to size the buffer. */
int agz_mcts_to_dot(agz_mcts* mcts, int max_nodes, char* buf, size_t cap, size_t* needed);
/* mcts.Config.Timeout (mcts/tree.go:18,34) — the reference's OWN stopping rule: its Search runs simulations until the wall clock
 * says stop (search.go:132-133,196-197; Budget is inert there), 100 ms in mcts.DefaultConfig.  Opt-in: timeout_ms > 0
 * makes agz_mcts_search run simulations for that long (the clock is read between slices of device work; a Budget > 0 still caps the
 * search), timeout_ms = 0 (default) restores the deterministic "exactly Budget simulations".  NOT deterministic: the simulation count
 * depends on the machine — parity tests use Budget.  With Budget <= 0 (what the Go shim passes for a reference conf that only sets
 * Timeout) agz_mcts_create sizes the node pool for 65536 expansions when max_nodes is 0, and a pool that fills up ENDS the search
 * (AGZ_OK, the best move of the tree as it stands) instead of failing with AGZ_E_TREE_FULL: the reference's arena is unbounded and only
 * its clock stops it.  agz_mcts_last_simulations: simulations the last agz_mcts_search ran. */
int agz_mcts_set_timeout_ms(agz_mcts* mcts, int timeout_ms);
int agz_mcts_last_simulations(agz_mcts* mcts, int64_t* sims);
/* (*MCTS).Nodes() (tree.go:126): nodes of the live tree (the reference counts its arena slots, freed ones included) */
int agz_mcts_nodes(agz_mcts* mcts, int* n_nodes);
int agz_mcts_get_stats(agz_mcts* mcts, agz_arena_stats* out);
/* (*MCTS).Reset() (tree.go:249-276) completed to what Arena.Play does with it — Reset then a fresh mcts.New
 * (arena.go:140-141,175-176; the reference's Reset alone leaves an unusable tree): empty tree, empty policy cache */
int agz_mcts_reset(agz_mcts* mcts);

/* ---- example sets on device: what sits between AZ.SelfPlay and dual.Train ----------------------
 * []agogo.Example (datatypes.go:41-46) as three device arrays: Board [n, F*H*W], Policy [n, PolicyLen], Value [n].
 * The payload (27.4 KB per 19x19 example) never leaves HBM; the host handles 4-byte row indices only. */
typedef struct agz_examples agz_examples;
int agz_examples_create(agz_ctx* ctx, int Features, int Height, int Width, int PolicyLen, agz_examples** out);
void agz_examples_destroy(agz_examples* ex);
int agz_examples_count(const agz_examples* ex, int64_t* n);
int agz_examples_clear(agz_examples* ex);
/* `ex = append(ex, a.SelfPlay()...)` (agogo.go:110-114) for all FINISHED games of a batched arena, in the reference's order:
 * game after game, each in ply order. Device-to-device.  The appended rows are TAKEN out of the arena
 * (agz_arena_drop_labelled_examples): calling it again appends only games that finished since. */
int agz_examples_append_arena(agz_examples* ex, agz_arena* arena);
/* append n rows from device buffers (e.g. the RCCL all-gathered examples of the other ranks) / host buffers */
int agz_examples_append_dev(agz_examples* ex, const float* planes_dev, const float* policy_dev, const float* value_dev, int64_t n);
int agz_examples_append_host(agz_examples* ex, const float* planes, const float* policy, const float* value, int64_t n);
/* device pointers of the raw store (rows in append order) — the send buffers of the RCCL all-gather */
int agz_examples_raw_dev(agz_examples* ex, float** planes, float** policy, float** value);
/* read back (tests / host-side consumers). *n = rows held; copies min(cap, *n) rows. */
int agz_examples_get(agz_examples* ex, float* planes, float* policy, float* value, int64_t cap, int64_t* n);
/* An Augmenter (datatypes.go:38-39; applied per recorded example, arena.go:115-120) built on RotateBoard
 * (encoding_helper.go:80-107): every example e is replaced by [e, rot e, rot^2 e, rot^3 e] — every plane of Board
 * and the m*n board part of Policy rotated, the pass entry and Value kept.  Non-square boards: AGZ_E_INVALID with
 * RotateBoard's message. */
int agz_examples_augment_rotate(agz_examples* ex);
/* `if maxExamples > 0 && len(ex) > maxExamples { shuffleExamples(ex); ex = ex[:maxExamples] }` (agogo.go:118-121;
 * maxExamples <= 0: skipped) followed by prepareExamples (agogo.go:211-249): shuffleExamples (agogo.go:251-257, build
 * RNG), batches = len/BatchSize, the first batches*BatchSize rows tensorised.  *batches may be 0 ("batches is nil",
 * agogo.go:123-125 — the caller's error). */
int agz_examples_prepare(agz_examples* ex, int BatchSize, int maxExamples, uint64_t seed, int* batches);
/* the prepared tensors: device pointers for agz_train_dev / host copies */
int agz_examples_tensors_dev(agz_examples* ex, float** Xs, float** Policies, float** Values, int64_t* rows, int* batches);
int agz_examples_get_tensors(agz_examples* ex, float* Xs, float* Policies, float* Values);
/* RotateBoard (encoding_helper.go:80-107) on `count` boards of m x n floats (host buffers; runs on the device). */
int agz_rotate_boards(agz_ctx* ctx, const float* boards, int count, int m, int n, float* out);

/* ---- multi-GPU exchange over RCCL / xGMI -------------------------------------------------------------------
 * Self-play games shard across GPUs with NO data-path collective (one agz_ctx + arena set per GPU).  The path has exactly
 * two exchange steps, both around dual.Train: the union of all ranks' examples (agogo.go:110-133: `ex` holds every episode
 * before shuffleExamples / prepareExamples) and the gradient sum of the data-parallel training step (dualnet/meta.go:33-40).
 * librccl is loaded on first use; without it these calls fail with AGZ_E_UNSUPPORTED (no single-GPU fallback). */
typedef struct agz_comm agz_comm;
#define AGZ_COMM_ID_BYTES 128
/* one process driving n GPUs (a Go host: one goroutine locked to an OS thread per ctx): ncclCommInitAll over the ctxs' devices;
 * comms[i] belongs to ctxs[i].  Collective calls on the n communicators must be issued concurrently, one thread per ctx. */
int agz_comm_init_all(agz_ctx* const* ctxs, int n, agz_comm** comms);
/* one process per GPU: rank 0 makes the id (agz_comm_unique_id), the host ships its AGZ_COMM_ID_BYTES bytes to the other ranks */
int agz_comm_unique_id(void* id128);
int agz_comm_init_rank(agz_ctx* ctx, int n_ranks, int rank, const void* id128, agz_comm** out);
void agz_comm_destroy(agz_comm* comm);
int agz_comm_rank(const agz_comm* comm);
int agz_comm_size(const agz_comm* comm);
/* every rank's example set becomes the union of all ranks' sets, in rank order (each rank's rows keep their order) */
int agz_examples_allgather(agz_comm* comm, agz_examples* ex);
/* sum the flat gradient buffer of the trainer over all ranks (one collective per step, in place, asynchronous on the ctx
 * stream); follow with agz_trainer_apply(t, lr, 1.0f / agz_comm_size(comm)) */
int agz_trainer_allreduce(agz_comm* comm, agz_trainer* t);
/* The data-parallel step with the reduction UNDER the backward pass: agz_trainer_forward_backward (host buffers) /
 * agz_trainer_forward_backward_dev (device buffers) on this rank's batch, every slice of the flat gradient buffer — the heads, then
 * layer L .. 0 as the backward pass finishes them — summed over the ranks on the communicator's own queue while the rest of the
 * backward runs (the G19 trainer's buffer is 7.7 GB: as one call after the backward it would cost about as much xGMI time as the
 * whole compute step).  On return (asynchronous on the ctx stream, like agz_trainer_allreduce) the gradients are the sums; the result
 * equals forward_backward + agz_trainer_allreduce.  Follow with agz_trainer_apply(t, lr, 1.0f / agz_comm_size(comm)).  Every rank
 * must call it for the same step (the slices are collectives, issued in the same order everywhere).  Errors: a rank that fails part-way
 * still enters every collective of the step, and the step ends with a one-word status exchange — the call then fails on EVERY rank
 * (AGZ_E_PEER on the ranks that were fine themselves) and the gradients must not be applied; an error of RCCL itself is fatal for the
 * process group (destroy the communicator).  dualnet/meta.go:16-54. */
int agz_trainer_forward_backward_allreduce(agz_comm* comm, agz_trainer* t, const float* planes, const float* pi, const float* v, float* cost);
int agz_trainer_forward_backward_allreduce_dev(agz_comm* comm, agz_trainer* t, const float* planes_dev, const float* pi_dev, const float* v_dev, float* cost);

#ifdef __cplusplus
}
#endif
#endif /* AGZ_H */
