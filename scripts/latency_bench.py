"""BASELINE.json configs[4]: 19x19 Go inference-only Agent.Search (tournament shape) — ONE game, K=256, 40 dual
blocks, 1600 simulations per move; reports p50 / p90 move latency and the per-evaluation network latency.

A move = agz_arena_begin_move + agz_arena_simulate(Budget) + agz_arena_end_move (+ stream sync), i.e. what
Agent.Search (agent.go:76-81 -> mcts/search.go:92) costs the caller.  Batch is 1 (one tree), so the tower runs the
split-K path (net.hip launch_conv).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import agogo_amd as A
from agogo_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--K", type=int, default=256)
ap.add_argument("--L", type=int, default=40)
ap.add_argument("--size", type=int, default=19)
ap.add_argument("--sims", type=int, default=1600)
ap.add_argument("--moves", type=int, default=12)
ap.add_argument("--games", type=int, default=1)
ap.add_argument("--lanes", type=int, default=1, help="agz_arena_set_parallel: simulations per tree and round")
ap.add_argument("--compute", choices=["auto", "f32", "bf16x3", "wino", "wino_h2"], default="auto",
                help="wino_h2: AGZ_COMPUTE_WINO_H2 | AGZ_COMPUTE_FORCE (the Winograd fp16x2 tower at every batch size: lane rounds); "
                     "wino: AGZ_COMPUTE_WINO")
ap.add_argument("--open", type=int, default=0, help="random opening moves before the timed moves (mid-game trees)")
args = ap.parse_args()

ctx = A.Ctx(0)
S = args.size
net = A.Net(ctx, args.K, args.L, 2 * args.K, S, S, 18, S * S + 1, BatchSize=args.games, bn_mode=capi.BN_IDENTITY)
net.init_random(1337)
net.commit()
if args.compute == "auto":     # the latency regime's fp16x2 one-launch-per-layer kernel (f32: exact fp32 products, split-K)
    net.set_compute_mode(capi.COMPUTE_AUTO)
if args.compute == "bf16x3":   # not forced: one board is the latency regime -> split-K on the bf16 pipe
    net.set_compute_mode(capi.COMPUTE_BF16X3)
if args.compute == "wino":
    net.set_compute_mode(capi.COMPUTE_WINO)
if args.compute == "wino_h2":
    net.set_compute_mode(capi.COMPUTE_WINO_H2 | capi.COMPUTE_FORCE)
arena = A.Arena(ctx, capi.GAME_WQ, S, S, komi=7.5, encoder=capi.ENC_WQ, n_games=args.games, seed=7, Budget=args.sims,
                max_moves=S * S * 2)
arena.set_inferencer(0, capi.INF_NET, net)
arena.set_inferencer(1, capi.INF_NET, net)
arena.set_parallel(args.lanes)
arena.reset()
if args.open:
    arena.random_moves(np.full(args.games, args.open, np.int32), 1337)
lat = []
for mv in range(args.moves + 3):
    t0 = time.perf_counter()
    if mv == args.moves + 2:
        ctx.prof_enable(True)   # one extra, untimed move for the per-kernel breakdown (event records cost time)
    arena.begin_move()
    arena.simulate(args.sims)
    arena.end_move(False)
    ctx.sync()
    if 2 <= mv < args.moves + 2:
        lat.append(time.perf_counter() - t0)
ctx.prof_enable(False)
n_conv, ms_conv = ctx.prof_read(capi.PROF_CONV)
n_init, ms_init = ctx.prof_read(capi.PROF_CONV_INIT)
n_head, ms_head = ctx.prof_read(capi.PROF_HEADS)
n_sel, ms_sel = ctx.prof_read(capi.PROF_SELECT)
n_exp, ms_exp = ctx.prof_read(capi.PROF_EXPAND)
lat = np.array(lat)
print(json.dumps({
    "workload": f"{S}x{S} wq Agent.Search, K={args.K}, L={args.L}, {args.sims} sims/move, {args.games} tree(s)",
    "lanes": args.lanes, "compute": args.compute, "opening_moves": args.open, "moves_timed": len(lat), "p50_move_s": float(np.percentile(lat, 50)), "p90_move_s": float(np.percentile(lat, 90)),
    "ms_per_sim": float(np.median(lat)) / args.sims * 1e3,
    "dual_conv_ms": ms_conv / max(n_conv, 1), "init_conv_ms": ms_init / max(n_init, 1),
    "heads_ms": ms_head / max(n_head, 1), "select_ms": ms_sel / max(n_sel, 1), "expand_ms": ms_exp / max(n_exp, 1),
    "tower_ms_per_eval": (ms_conv + ms_init + ms_head) / max(n_head, 1),
}))
