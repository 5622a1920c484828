/*
 * agz_debug.h — measurement and test hooks of libagz.so.  NOT part of the drop-in surface (include/agz.h): nothing on the
 * reference side binds these; bench.py reads the kernel-class timers, the parity tests read single Winograd stages.
 */
#ifndef AGZ_DEBUG_H
#define AGZ_DEBUG_H

#include "agz.h"

#ifdef __cplusplus
extern "C" {
#endif

/* kernel-class timers: HIP events recorded on the ctx stream around every launch of a class.
 * enable=1 starts collecting (and clears), enable=0 stops.  agz_ctx_prof_read syncs and returns
 * the launch count and summed milliseconds of one class. */
#define AGZ_PROF_CONV 0    /* fused dual-branch 3x3 conv block launches (the dominant kernel) */
#define AGZ_PROF_HEADS 1   /* policy/value head kernel */
#define AGZ_PROF_SELECT 2  /* MCTS select/apply/encode */
#define AGZ_PROF_EXPAND 3  /* MCTS expand/backup */
#define AGZ_PROF_MOVE 4    /* root update / best move / apply */
#define AGZ_PROF_CONV_INIT 5 /* the single F->K input conv */
#define AGZ_PROF_WINO_IN 6   /* AGZ_COMPUTE_WINO: input transform; these three nest inside AGZ_PROF_CONV (the whole block) */
#define AGZ_PROF_WINO_GEMM 7 /* the 36 transform-domain GEMMs (the dominant kernel of that mode) */
#define AGZ_PROF_WINO_OUT 8  /* output transform + block epilogue */
#define AGZ_PROF_NCLASS 9
/* enable: 0 stop, 1 all classes, otherwise a mask with bit (k + 1) selecting class k (fewer event records per step) */
int agz_ctx_prof_enable(agz_ctx* ctx, int enable);
int agz_ctx_prof_read(agz_ctx* ctx, int klass, int64_t* launches, double* total_ms);
/* bracket only every stride-th launch of class klass (default 1): two event records per bracketed launch cost GPU time, and a
 * timed region that wants a kernel's average duration does not need every launch */
int agz_ctx_prof_set_stride(agz_ctx* ctx, int klass, int stride);

/* Diagnostics (tests): the first two stages of the Winograd path on host data.  x [B][H][W][C] (NHWC, C % 16 == 0),
 * w [N][C][3][3]  ->  V [36][T][C] = Bt d B of every 6x6 input tile, M [36][T][N] = V[pos] * (G g Gt)[pos],
 * T = B * ceil(H/4) * ceil(W/4) tiles in (board, tile row, tile column) order, pos = 6 * xi + nu. */
int agz_wino_stages(agz_ctx* ctx, const float* x, const float* w, int B, int H, int W, int C, int N, float* V, float* M);

/* Measurement: the Winograd tile size m (F(m x m, 3x3), m = 4 or 5) AGZ_COMPUTE_WINO_H2 uses on an H x W board — the one with
 * the fewest transform-domain rows, (m + 2)^2 * ceil(H/m) * ceil(W/m); bench.py prices the kernels' algorithmic bytes with it. */
int agz_wino_h2_tile(int H, int W);

/* Measurement: 1 when AGZ_COMPUTE_WINO_H2 runs the chained block (GEMM + fused out->in kernel, conv_wino_h2c.hpp) on an H x W board with K
 * filters, 0 when it runs the three-kernel block — bench.py prices the kernels' algorithmic bytes accordingly. */
int agz_wino_h2_chained(int H, int W, int K);

/* A/B hook: which form of the AGZ_COMPUTE_WINO_H2 block a net runs.  -1 (default): the chained form (output transform of block l
 * and input transform of block l+1 in one kernel, conv_wino_h2c.hpp) wherever the shape allows, else the three-kernel block;
 * 0: the three-kernel block; 1: as -1.  The environment switch AGZ_WINO_H2_FORM=0 (agz.h) does the same for a whole process. */
int agz_net_set_wino_h2_form(agz_net* net, int form);

/* A/B hook: which GEMM kernel the chained AGZ_COMPUTE_WINO_H2 block runs.  0 (default): the build's default; 1: wino_gemm_h2g_kernel
 * (128 x 256 tile, both operands streamed, three workgroups per CU); 2: wino_gemm_h2p_kernel (persistent, the weight slab stationary in
 * registers, M stores under the next tile's arithmetic; K = 256 only, other shapes keep kernel 1).  Results are bit-identical.  The
 * environment switch AGZ_WINO_H2_GEMM=1|2 (agz.h) does the same for a whole process.  Decomposition runs (timing only, the results are
 * NOT valid): 2 + 16 * mode with mode bit 0 = kernel 2 without its M stores, bit 1 = without its operand DMA (profiles/r05).
 * + 64 (with 0, 1 or 2): M and V2c stored with the default cache policy, as in round 4, instead of non-temporally (bit-identical; A/B). */
int agz_net_set_wino_h2_gemm(agz_net* net, int variant);

/* prepareRoot (mcts/search.go:392-408) evaluates the network only for roots without children.  agz_arena_begin_move packs those roots to
 * the front of the batch and runs the forward on the smallest batch that takes the same kernels as the arena's whole batch (per board the
 * results are bit-identical); on = 0 restores the whole-batch forward (A/B and parity hook).  agz_arena_last_prep_batch: boards the last
 * begin_move's forward ran on (0: skipped, no root needed it) and roots it had to evaluate. */
int agz_arena_set_prep_compact(agz_arena* arena, int on);
int agz_arena_last_prep_batch(agz_arena* arena, int* boards, int* roots);

/* A/B hook: the trainer's AGZ_COMPUTE_WINO_H2 forward convolutions through the DMA GEMM on pre-split fp16 planes (k_conv_h2dma, train.hip;
 * default on) or through conv3x3_h2w_kernel, which splits the fp32 activations while staging them (bit 0 of `on` clear).  Bit 2 of `on` set:
 * the first form of the head kernels (one thread per output, three-block BatchNorm passes) instead of the second (default).  Bit 3 set: every
 * layer's weight images (fp16 hi / lo filter image of the forward convolution, Winograd image of the data gradient) built per layer in line
 * instead of at the start of the step on the trainer's side stream (default).  Bit 4 set: no side stream at all
 * (diagnostic: per-kernel durations without overlap).  Bit 5 set: the DMA forward convolution with one tap per K step (k_conv_h2dma, 128 x 256
 * tile) instead of nine taps from one x image (k_conv_h2dma3, 256 x 128 tile; default where the image fits its 372 rows).  Same tolerance;
 * bits 3, 4 and 5 do not change a single product.  Bits 6, 7: decomposition runs of k_conv_h2dma3 (no x image after chunk 0 / no weight DMA):
 * WRONG results, timing only.  Bit 8 (round 6): the weight gradient k_wgrad_h2t3 at two workgroups per CU as in round 5 (default: ONE — it then leaves
 * half of every SIMD's registers to the main stream's chain, which runs beside it).  Bit 9: the side stream at the lowest priority (set before
 * the first step; measured no different).  Bits 8 and 9 do not change a single product. */
int agz_trainer_set_dma_forward(agz_trainer* t, int on);

/* The smallest batch >= n that runs the kernels of a batch of G boards on this net in its current compute mode (agz_net::min_same_batch, what
 * prepareRoot's packed forward uses): per board the outputs of such a batch are those of the G-board batch bit for bit.  The parity tests
 * evaluate single leaves for the oracle with it instead of G copies of the board. */
int agz_net_min_same_batch(agz_net* net, int n, int G, int* batch);

/* Measurement: raw device counter `which` of an arena.  AGZ_CNT_PATHMAX: nodes on the LONGEST descent since the last reset (all games run
 * one simulation per step in lock step: a step's k_select lasts as long as its longest path, so the maximum — not the mean that
 * agz_arena_stats reports — is what prices the kernel on narrow, deep trees). */
#define AGZ_CNT_PATHMAX 15
int agz_arena_debug_counter(agz_arena* arena, int which, int64_t* value);

/* Measurement: the node-pool capacity of an arena's trees and how often AGZ_POOL_GROW has re-allocated them */
int agz_arena_pool_capacity(agz_arena* arena, int* nodes_per_pool, int* grows);

/* Failure injection (tests): the data-parallel step of this communicator fails on THIS rank right before it would enter the collective of
 * slice k (0 = the heads, 1 .. = layer L .. 0; k < 0: off).  One shot: cleared by the step it hits.  What the test checks is the rule of
 * agz_trainer_forward_backward_allreduce: the failing rank still enters every collective, every rank's call fails (AGZ_E_PEER on the
 * healthy ones), nobody hangs, and the next step is a normal one. */
int agz_comm_debug_fail_slice(agz_comm* comm, int k);

#ifdef __cplusplus
}
#endif
#endif /* AGZ_DEBUG_H */
