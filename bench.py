#!/usr/bin/env python3
"""bench.py — MCTS simulations/sec of batched self-play on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

Workload (configs[3] per GPU, the configuration the metric is quoted on): 19x19 Go (game/wq), K=256,
20 dual-branch blocks, FC=512, ActionSpace 362, WQEncoder (F=18), 512 concurrent self-play games per GPU,
800 simulations per move, synthetic random-init weights (fixed seed), games started from the empty board.

One STEP = one simulation for every game on the GPU: PUCT descent + Apply in each game's tree (k_select),
ONE batched 512-leaf pass through the conv tower and heads, expansion + backup (k_expand).  Every
`Budget` steps the move is finished (bestMove, Apply, tree re-root) and the next one prepared — inside the
timed region.  value = non-null simulations (search.go:175-178) completed by all ranks / max-over-ranks time.

Steady state: before anything is timed every game gets its OWN position — u ~ U[0, floor(0.6*H*W)] uniformly
random legal moves (agz_arena_random_moves, deterministic per slot) — one whole move is searched (its wall time
is the measured 19x19 moves/s in `extra`), and the next move's tree is grown to Budget - warmup - steps/2 simulations, so the
timed steps run on deep trees and straddle a move boundary: end_move (bestMove, example record, Apply, Ended) and begin_move
(tree re-root, prepareRoot) of all games are inside the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import agogo_amd as A  # noqa: E402
from agogo_amd import capi  # noqa: E402
from agogo_amd import dist as adist  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2516.6  # same guide: v_mfma_f32_32x32x16_bf16 dense (256 CU x 4 SIMD x 1024 FLOP/clk x 2.4 GHz)
# bf16x3 formulation (agogo_amd/csrc/conv_x3.hpp): 6 bf16 MFMAs per fp32-grade product -> algorithmic peak = bf16 peak / 6
BF16X3_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0
# fp16x2 formulation (conv_h2.hpp): 3 fp16 MFMAs per product (fp16 dense peak = bf16 dense peak)
FP16X2_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3.0
MODES = {"f32": capi.COMPUTE_F32_MFMA, "bf16x3": capi.COMPUTE_BF16X3, "fp16x2": capi.COMPUTE_FP16X2, "wino": capi.COMPUTE_WINO,
         "wino_h2": capi.COMPUTE_WINO_H2}
HBM_PEAK_GBS = 8000.0          # same guide: HBM3E ~8 TB/s
N_ONE_QUEUE_STEPS = 24         # bracketed one-queue steps after the timed region: the roofline's per-kernel durations (round 5: 6 — 120 launches, a noisy sample)


def standard_bn_init(net):
    """gamma = 1, beta = 0 (identity statistics): keeps random-init activations O(1) so the synthetic search trees
    branch like a trained net's instead of saturating softmax/tanh (DESIGN.md §synthetic inputs)."""
    for i in range(net.num_params()):
        name, n = net.param_info(i)
        if name.endswith("_gamma"):
            net.set_param(i, np.ones(n, np.float32))
        elif name.endswith("_beta"):
            net.set_param(i, np.zeros(n, np.float32))


def _oracle_net(O, K, L, FC, W, H, F, A_):
    net = O.Net(K, L, FC, W, H, F, A_, bn_mode=2)
    net.init_random(1337)
    for i in range(net.num_params()):
        nm = net.param_name(i)
        if nm.endswith("_gamma"):
            net.set_param(i, np.ones_like(net.get_param(i)))
        elif nm.endswith("_beta"):
            net.set_param(i, np.zeros_like(net.get_param(i)))
    return net


def _oracle_selfplay_leg(O, T, kind, m, n, k, komi, enc, net, budget, moves_per_thread, opening=None, complete=False, max_moves=0):
    """T oracle arenas, one thread each (the oracle calls run outside the GIL): every thread plays `moves_per_thread` searched moves
    (complete=True: whole games until Ended) of its own game.  Returns measured sims/s, evals/s, moves/s, games finished."""
    import threading
    arenas = []
    rng = np.random.default_rng(1337)
    for g in range(T):
        ar = O.Arena(kind, m, n, k, komi=komi, enc=enc, Budget=budget, seed=1337 + g, max_moves=max_moves)
        ar.set_inferencer(0, O.INF_NET, net)     # the net is read-only during inference
        ar.set_inferencer(1, O.INF_NET, net)
        ar.begin(g % 2)
        if opening:
            for _ in range(int(rng.integers(0, opening + 1))):
                ar.random_move(1337, g)
        arenas.append(ar)
    moves = [0] * T
    done = [0] * T

    def work(i):
        ar = arenas[i]
        while True:
            alive = ar.step(True)
            moves[i] += 1
            if not alive:
                done[i] += 1
                break
            if not complete and moves[i] >= moves_per_thread:
                break

    threads = [threading.Thread(target=work, args=(i,)) for i in range(T)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt = time.perf_counter() - t0
    stats = [[ar.tree_stats(a) for a in (0, 1)] for ar in arenas]
    playouts = sum(st["playouts"] for pair in stats for st in pair)
    evals = sum(st["nn_evals"] for pair in stats for st in pair)
    return {"sims_per_s": playouts / dt, "evals_per_s": evals / dt, "moves_per_s": sum(moves) / dt, "seconds": dt, "threads": T,
            "sims": playouts, "moves": sum(moves), "games_finished": sum(done), "games_per_s": (sum(done) / dt) if complete else None}


def cpu_baseline(size, K, L, budget_s=30.0, max_threads=64, g19_seconds=30.0):
    """The oracle (CPU restatement of the reference algorithm: per-leaf inference, sequential pipeline) timed on the box's host
    cores, SURVEY 8(d) / BASELINE.md section 3: one independent game per thread (the way the reference would use its cores), threads
    = min(host cores, 64).  Legs, ~30 s in total:
      ttt           mnk.TicTacToe(), dual.DefaultConf(3, 3, 10), Budget 1000: COMPLETE games (configs[0])
      c4            Connect-4 6x7, K=64, 6 blocks, 400 sims/move: two searched moves per thread (a complete game is ~1.5 min of CPU)
      go9           9x9 Go, K=128, 10 blocks: one searched move of 24 sims per thread (400 sims/move = ~1 min of CPU per move)
      g19_fair      19x19, K=256, 20 blocks, batch 1 per leaf: one move of a few sims per thread from a random mid-game opening
      g19_faithful  the reference's Inferencer.Infer evaluates an ActionSpace-row batch per leaf (dualnet/meta.go:175-189): 362 rows;
                    timed: one row per thread, all threads at once -> seconds per 362-row leaf on these cores
    `value` is the g19_fair sims/s (the configuration the metric is quoted on)."""
    import threading
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    cores = os.cpu_count() or 2
    # SURVEY 8(d) asks for T = hardware_concurrency.  Measured in round 4 on the GPU box (profiles/r04/cpu_baseline_all_cores.json): with all
    # 256 host threads the memory-bound oracle is SLOWER in aggregate (g19_fair 2.8 sims/s against 5.4 at 64 threads, go9 138 against 245,
    # c4 1.7 k against 3.2 k) and the legs take seven minutes — so the default stays min(cores, 64), the faster configuration, and
    # --cpu-threads 0 runs every core
    T = max(1, min(max_threads, cores) if max_threads else cores)
    legs = {}
    # --- ttt: complete games
    net = _oracle_net(O, 3, 3, 6, 3, 3, 2, 10)
    r = _oracle_selfplay_leg(O, T, O.MNK, 3, 3, 3, 0.0, O.ENC_TWOPLANE, net, 1000, 0, complete=True)
    legs["ttt"] = dict(r, workload="mnk.TicTacToe(), dual.DefaultConf(3,3,10) (K=3, 3 blocks), Budget 1000, %d complete games" % T)
    # --- c4
    net = _oracle_net(O, 64, 6, 128, 7, 6, 2, 8)
    r = _oracle_selfplay_leg(O, T, O.C4, 6, 7, 4, 0.0, O.ENC_TWOPLANE, net, 400, 2)
    legs["c4"] = dict(r, workload="Connect-4 6x7, K=64, 6 blocks, 400 sims/move: %d games x 2 searched moves" % T,
                      games_per_s_derived=r["moves_per_s"] / 21.0, derived_note="moves/s / 21 moves per game (what the GPU leg's games take)")
    # --- go9
    net = _oracle_net(O, 128, 10, 256, 9, 9, 18, 82)
    r = _oracle_selfplay_leg(O, T, O.WQ, 9, 9, 0, 7.5, O.ENC_WQ, net, 24, 1)
    legs["go9"] = dict(r, workload="9x9 Go (wq), K=128, 10 blocks: %d games x 1 searched move of 24 sims" % T,
                       games_per_s_derived=r["sims_per_s"] / (400.0 * 64.0), derived_note="sims/s / (400 sims x 64 moves per game)")
    # --- g19 fair
    A_ = size * size + 1
    net = _oracle_net(O, K, L, 2 * K, size, size, 18, A_)
    x = np.zeros((1, 18, size, size), np.float32)
    t0 = time.perf_counter()
    net.infer(x)
    t_eval = time.perf_counter() - t0
    # under T concurrent evaluations one evaluation takes ~4.5x its solo time at 64 threads on this class of host (memory
    # bandwidth), more with every core busy
    slow = 4.5 * max(1.0, T / 64.0) ** 0.5   # (measured in round 6: 5.9 s per evaluation under 64 concurrent ones against 1.3 s solo)
    # (round 5 sized this leg for ~9 s: 2 simulations per thread, 128 in all, and the figure moved 5.2 - 7.9 sims/s between rounds on the same
    # code; now a ~30 s box, at least 4 simulations per thread — VERDICT r5 weak 10)
    sims = int(max(4, min(64, g19_seconds / max(slow * t_eval, 1e-3) - 1)))
    r = _oracle_selfplay_leg(O, T, O.WQ, size, size, 0, 7.5, O.ENC_WQ, net, sims, 1, opening=int(0.6 * size * size))
    legs["g19_fair"] = dict(r, workload="19x19 Go, K=%d, %d blocks, batch 1 per leaf: %d games from random mid-game openings x 1 move of %d sims"
                            % (K, L, T, sims), solo_eval_seconds=t_eval)
    # --- g19 faithful: one row per thread, T rows at once; a leaf is A_ rows
    xs = [np.zeros((1, 18, size, size), np.float32) for _ in range(T)]
    threads = [threading.Thread(target=net.infer, args=(xs[i],)) for i in range(T)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt_rows = time.perf_counter() - t0
    leaf_s = dt_rows * (A_ / float(T))
    legs["g19_faithful"] = {"rows_timed": T, "seconds": dt_rows, "rows_per_leaf": A_, "seconds_per_leaf": leaf_s, "sims_per_s": 1.0 / leaf_s,
                            "threads": T,
                            "workload": "the reference evaluates an ActionSpace-row batch per leaf and keeps row 0 (dualnet/meta.go:175-189): "
                                        "%d of a leaf's %d rows timed (one per thread, concurrently); seconds_per_leaf = that time x %d / %d — "
                                        "a whole leaf on all %d threads" % (T, A_, A_, T, T)}
    total = sum(v["seconds"] for v in legs.values())
    return {"value": legs["g19_fair"]["sims_per_s"], "unit": "sims/s", "cores": T, "kind": "port",
            "sample": "oracle (C++ restatement, per-leaf inference), %d threads on the box's %d host cores, one game per thread; legs ttt / c4 / go9 / "
                      "g19_fair / g19_faithful, %.1f s in total; value = g19_fair (19x19, K=%d, %d blocks: %d sims in %.1f s)"
                      % (T, cores, total, K, L, legs["g19_fair"]["sims"], legs["g19_fair"]["seconds"]),
            "method": "threads = min(host cores, 64) as in round 3 (round 2: min(32, cores/2); round 1: min(32, cores)); every host core (256) was "
                      "measured in round 4 and is slower in aggregate (g19_fair 2.8 sims/s, profiles/r04/cpu_baseline_all_cores.json); "
                      "per_core_sims_per_s is the figure comparable across thread counts",
            "per_core_sims_per_s": legs["g19_fair"]["sims_per_s"] / T, "evals_per_s": legs["g19_fair"]["evals_per_s"],
            "host_cores": cores, **legs}


def games_leg(ctx, compute="bf16x3"):
    """Measured games/s (not an estimate) on BASELINE config #2: Connect-4, K=64, 6 blocks, 256 concurrent games,
    400 sims/move, continuous self-play until 256 games have finished."""
    net = A.Net(ctx, 64, 6, 128, 7, 6, 2, 8, bn_mode=capi.BN_IDENTITY)
    net.init_random(1337)
    standard_bn_init(net)
    net.commit()
    net.set_compute_mode(MODES[compute])
    # RandomCount = 4 (mcts.Config.RandomCount, tree.go:22; search.go:356): the first four moves of every game are drawn from the visit
    # distribution by the tree's own RNG (one stream per tree) — the reference's mechanism for game diversity.  With RandomCount = 0
    # (round 5) the deterministic search made all 256 games ONE game.
    arena = A.Arena(ctx, capi.GAME_C4, 6, 7, 4, encoder=capi.ENC_TWOPLANE, n_games=256, seed=1337, Budget=400, RandomCount=4,
                    RandomMinVisits=1, RandomTemperature=1.0)
    arena.set_inferencer(0, capi.INF_NET, net)
    arena.set_inferencer(1, capi.INF_NET, net)
    arena.reset()
    ctx.sync()
    t0 = time.perf_counter()
    arena.play(0, record=True)          # every one of the 256 games once, to its end (no restarts: an unbiased length sample)
    ctx.sync()
    dt = time.perf_counter() - t0
    st = arena.stats()
    lens = np.array([len(arena.history(g)) for g in range(256)])
    out = {"workload": "config #2: Connect-4 6x7, K=64, 6 blocks, 256 concurrent games, 400 sims/move, RandomCount 4: each game played once to its end",
           "games_finished": st["games_finished"], "seconds": dt, "games_per_s": st["games_finished"] / dt,
           "sims_per_s": st["sims_nonnull"] / dt, "moves_per_s": st["moves_played"] / dt, "examples": st["examples"],
           "game_length": _length_stats(lens), "distinct_games": len({arena.history(g).tobytes() for g in range(256)}),
           **_continuous_rate(256, dt, lens),
           "note": "the arena runs until its LONGEST game ends (finished games idle in the batch): games_per_s is a lower bound of the continuous-self-play rate"}
    arena.close()
    net.close()
    return out


def _length_stats(lens):
    lens = np.asarray(lens)
    return {"mean": float(lens.mean()), "min": int(lens.min()), "p10": float(np.percentile(lens, 10)), "p50": float(np.percentile(lens, 50)),
            "p90": float(np.percentile(lens, 90)), "max": int(lens.max()), "distinct_lengths": int(len(np.unique(lens)))}


def _continuous_rate(G, seconds, lens):
    """continuous self-play (finished games restart at once, agz_arena_selfplay) keeps all G slots busy: one arena ply costs the same
    whether a slot holds a live game or not (the batch is fixed), so its rate is G games per (mean length x seconds per arena ply)"""
    lens = np.asarray(lens)
    per_ply = seconds / max(1, int(lens.max()))
    return {"seconds_per_arena_ply": per_ply, "games_per_s_continuous": G / (float(lens.mean()) * per_ply),
            "games_per_s_continuous_note": "G / (mean game length x seconds per arena ply): the rate with every slot restarted as it finishes; "
                                           "both factors measured in this run"}


def go9_leg(ctx, compute="wino_h2"):
    """BASELINE config #3: 9x9 Go (wq), K=128, 10 blocks, 512 concurrent games, 400 sims/move (wino_h2 with F(5x5,3x3): 4 tiles x 49 positions
    per board, measured 15 % less time per 512-board pass than bf16x3 on this shape).  Two measurements, as for 19x19: (1) the MOVE RATE at the
    configuration's own 400 simulations per move — 16 whole arena plies of all 512 games, each game standing at a uniformly drawn ply of a
    game it really played; (2) the GAME LENGTH — every one of 512 games played once to its end at 16 simulations per move (RandomCount 8: the
    first eight moves drawn from the visit distribution by each tree's own RNG, the reference's mechanism for game diversity: tree.go:22,
    search.go:356).  games/s = (1) / mean of (2).
    Round 5 played 512 COPIES of one deterministic game (64 moves); complete games at 400 simulations per move were measured once this round
    (99 s: mean 131 moves, 340 of 512 at the 162-move cap, 5.1 games/s as a batch, 6.4 continuous: profiles/r06/go9_complete_games_400_sims.json)."""
    K, L, G, sims = 128, 10, 512, 400
    net = A.Net(ctx, K, L, 2 * K, 9, 9, 18, 82, bn_mode=capi.BN_IDENTITY)
    net.init_random(1337)
    standard_bn_init(net)
    net.commit()
    net.set_compute_mode(MODES[compute])
    # (2) first: the GAME LENGTH — and the games themselves, whose positions (1) is then measured on
    arena = A.Arena(ctx, capi.GAME_WQ, 9, 9, 0, 7.5, encoder=capi.ENC_WQ, n_games=G, seed=1337, Budget=16, RandomCount=8,
                    RandomMinVisits=1, RandomTemperature=1.0)
    arena.set_inferencer(0, capi.INF_NET, net)
    arena.set_inferencer(1, capi.INF_NET, net)
    arena.reset()
    ctx.sync()
    t0 = time.perf_counter()
    arena.play(0, record=True)
    ctx.sync()
    dl = time.perf_counter() - t0
    st = arena.stats()
    hist = [arena.history(g) for g in range(G)]
    lens = np.array([len(h) for h in hist])
    complete = {"games_finished": st["games_finished"], "seconds": dl, "game_length": _length_stats(lens),
                "termination": _termination_mix(arena, G, 2 * 81),
                "distinct_games": len({h.tobytes() for h in hist}), "examples": st["examples"],
                "examples_dropped": st["examples_dropped"], "tree_full": st["tree_full"]}
    arena.close()
    # (1) move rate at 400 simulations per move, on positions of THOSE games: game g replayed up to a uniformly drawn ply of its own length (an
    # arena ply late in a game costs ~1.5x an early one; uniformly random openings — round 6's first forms of this leg — are neither: 1284 and
    # 533 moves/s where complete 400-simulation games ran 834).  Node pools: the default with AGZ_POOL_GROW (round 5's default of two searches'
    # worth was outgrown by one of 512 different games in round 6's first run)
    arena = A.Arena(ctx, capi.GAME_WQ, 9, 9, 0, 7.5, encoder=capi.ENC_WQ, n_games=G, seed=1337, Budget=sims)
    arena.set_inferencer(0, capi.INF_NET, net)
    arena.set_inferencer(1, capi.INF_NET, net)
    arena.set_pool_policy(capi.POOL_GROW)     # default pools (four searches' worth) that grow if a narrow tree needs more
    arena.reset(np.array([1] * G, np.uint8))
    plies = 16
    prefix = np.array([int(np.random.default_rng(1337 + g).integers(0, max(1, len(hist[g]) - plies - 1))) for g in range(G)])
    for ply in range(int(prefix.max())):
        mv = np.array([int(hist[g][ply]) if ply < prefix[g] else capi.NO_MOVE for g in range(G)], np.int32)
        arena.apply_moves(mv)
    arena.play(1, record=True)            # (first ply: every root fresh)
    ctx.sync()
    s0 = arena.stats()
    t0 = time.perf_counter()
    arena.play(plies, record=True)
    ctx.sync()
    dt = time.perf_counter() - t0
    s1 = arena.stats()
    rate = {"plies": plies, "seconds": dt, "moves_per_s": (s1["moves_played"] - s0["moves_played"]) / dt,
            "sims_per_s": (s1["sims_nonnull"] - s0["sims_nonnull"]) / dt, "active_games_at_end": s1["n_active"], "tree_full": s1["tree_full"],
            "positions": "each game replayed to a uniformly drawn ply of its own (16-simulation) game: mean %.0f moves played" % float(prefix.mean()),
            "node_pool": dict(zip(("nodes_per_pool", "grows"), arena.pool_capacity()))}
    arena.close()
    out = {"workload": "config #3: 9x9 Go (wq, komi 7.5), K=128, 10 blocks, 512 concurrent games, 400 sims/move", "compute": compute,
           "move_rate_at_400_sims": rate,
           "complete_games_at_16_sims": complete,
           "games_per_s": rate["moves_per_s"] / float(lens.mean()), "sims_per_s": rate["sims_per_s"], "moves_per_s": rate["moves_per_s"],
           "moves_per_game": float(lens.mean()),
           "games_per_s_note": "moves/s of 16 arena plies at 400 simulations per move (512 games on positions of games they really played; the few that end inside the window stop counting) / the mean length of 512 complete games at 16 "
                               "simulations per move; complete games AT 400 simulations: profiles/r06/go9_complete_games_400_sims.json (mean 131 moves)"}
    net.close()
    return out


def _termination_mix(arena, G, cap):
    """how the games of an arena ended: two passes in a row, a resignation, or the move cap"""
    two_pass = resign = capped = other = 0
    for g in range(G):
        h = arena.history(g)
        if len(h) >= 1 and h[-1] == capi.RESIGN:
            resign += 1
        elif len(h) >= 2 and h[-1] == capi.PASS and h[-2] == capi.PASS:
            two_pass += 1
        elif len(h) >= cap:
            capped += 1
        else:
            other += 1
    return {"two_passes": two_pass, "resign": resign, "move_cap": capped, "other": other}


def complete_19x19_leg(ctx, net, G=512, budget=16, S=19):
    """512 COMPLETE 19x19 games, once, on the headline network (K=256, 20 blocks, AGZ_COMPUTE_WINO_H2) at a small Budget —
    Arena.Play to Ended() (arena.go:96-138) for every game of the arena: the measured game-length distribution and termination mix
    (two passes / resignation / the 2*M*N move cap) that the headline's games/s figure divides by.  A game at 800 simulations per move is
    hours; the LENGTH of a game under this network is what the small-Budget run measures (RandomCount 16: every game its own)."""
    arena = A.Arena(ctx, capi.GAME_WQ, S, S, komi=7.5, encoder=capi.ENC_WQ, n_games=G, seed=4242, Budget=budget, PUCT=1.0, RandomCount=16,
                    RandomMinVisits=1, RandomTemperature=1.0, DumbPass=True, PassPreference=capi.DONT_PREFER_PASS)
    arena.set_inferencer(0, capi.INF_NET, net)
    arena.set_inferencer(1, capi.INF_NET, net)
    arena.reset()
    ctx.sync()
    t0 = time.perf_counter()
    arena.play(0, record=True)
    ctx.sync()
    dt = time.perf_counter() - t0
    st = arena.stats()
    lens = np.array([len(arena.history(g)) for g in range(G)])
    res = arena.results()
    out = {"workload": "%d complete 19x19 games (each once, to Ended()), K=256, 20 blocks, %d sims/move, RandomCount 16" % (G, budget),
           "games_finished": st["games_finished"], "seconds": dt, "moves_played": st["moves_played"], "sims": st["sims_nonnull"],
           "game_length": _length_stats(lens), "termination": _termination_mix(arena, G, 2 * S * S),
           "distinct_games": len({arena.history(g).tobytes() for g in range(G)}),
           "results": res, "examples": st["examples"], "examples_dropped": st["examples_dropped"],
           "tree_full": st["tree_full"]}
    arena.close()
    return out


def _opening_planes(ctx, S, n=64, seed=1337):
    """n encoded mid-game positions (WQEncoder planes [n, 18, S, S]) through the product path: a small arena on random openings plays one
    recorded move with the synthetic hash inferencer; the example rows are the encoder's tensors of those positions"""
    ar = A.Arena(ctx, capi.GAME_WQ, S, S, komi=7.5, encoder=capi.ENC_WQ, n_games=n, seed=seed, Budget=2)
    ar.set_inferencer(0, capi.INF_HASH)
    ar.set_inferencer(1, capi.INF_HASH)
    ar.reset()
    ar.random_moves(np.random.default_rng(seed).integers(8, int(0.6 * S * S) + 1, size=n).astype(np.int32), seed)
    ar.begin_move(); ar.simulate(2); ar.end_move(True)
    planes = ar.examples()[0]
    ar.close()
    return planes.reshape(-1, 18, S, S)


def peaked_net(ctx, S, K, L, compute, target_max_prior, zero_value=False):
    """The headline network with a PEAKED policy: same random-init tower (seed 1337, gamma = 1, beta = 0, identity statistics),
    the policy FC scaled by c so that the mean largest prior over 64 mid-game positions is `target_max_prior`.  The FC bias is zero at
    initialisation (ermahagerdmonards.go:82), so the logits scale with c exactly and p_c is proportional to p_1 ** c: c is solved on the host
    from one evaluation and checked with a second one after the commit.  zero_value: the value output layer zeroed (every evaluation 0, Q equal
    for all children of a node): the search then follows the priors alone — the narrowest, deepest trees this network family can produce."""
    Aspace = S * S + 1
    net = A.Net(ctx, K, L, 2 * K, S, S, 18, Aspace, bn_mode=capi.BN_IDENTITY)
    net.init_random(1337)
    standard_bn_init(net)
    net.commit()
    net.set_compute_mode(MODES[compute])
    x = _opening_planes(ctx, S)
    p1, _ = net.infer(x)
    lp = np.log(np.maximum(p1.astype(np.float64), 1e-300))

    def mean_max(c):
        z = c * lp
        z -= z.max(axis=1, keepdims=True)
        e = np.exp(z)
        return float((e / e.sum(axis=1, keepdims=True)).max(axis=1).mean())

    lo, hi = 1.0, 1.0
    while mean_max(hi) < target_max_prior and hi < 1e6:
        hi *= 2.0
    for _ in range(50):
        mid = 0.5 * (lo + hi)
        if mean_max(mid) < target_max_prior:
            lo = mid
        else:
            hi = mid
    c = hi
    for i in range(net.num_params()):
        name, cnt = net.param_info(i)
        if name == "Policy_w":
            net.set_param(i, (net.get_param(i).astype(np.float64) * c).astype(np.float32))
        elif zero_value and name == "ValueOutput_w":
            net.set_param(i, np.zeros(cnt, np.float32))
    net.commit()
    net.set_compute_mode(MODES[compute])
    p2, v2 = net.infer(x)
    return net, {"policy_fc_scale": c, "mean_max_prior": float(p2.max(axis=1).mean()), "mean_max_prior_before": float(p1.max(axis=1).mean()),
                 "mean_abs_value": float(np.abs(v2).mean())}


def deep_tree_leg(ctx, S, K, L, G, budget, compute, target_max_prior=0.5, zero_value=False, steps_prof=24):
    """the headline workload on NARROW, DEEP trees.  The headline's near-uniform priors (gamma = 1 / beta = 0) give ~250 children per
    visited node and paths of ~3 nodes; a trained policy is peaked.  Same arena shape, same tower, peaked policy head (peaked_net): one whole
    move of all games (begin_move + Budget simulations + end_move), its sims/s beside the headline's, the tree shape it ran on (mean path
    nodes, children read per level), and k_select / k_expand per step from HIP events on the LAST steps of the move (deepest trees)."""
    net, info = peaked_net(ctx, S, K, L, compute, target_max_prior, zero_value)
    arena = A.Arena(ctx, capi.GAME_WQ, S, S, komi=7.5, encoder=capi.ENC_WQ, n_games=G, seed=1337, Budget=budget, PUCT=1.0, RandomCount=0,
                    DumbPass=True, PassPreference=capi.DONT_PREFER_PASS)
    arena.set_inferencer(0, capi.INF_NET, net)
    arena.set_inferencer(1, capi.INF_NET, net)
    arena.reset()
    n_open = np.random.default_rng(1337).integers(0, int(0.6 * S * S) + 1, size=G).astype(np.int32)
    arena.random_moves(n_open, 1337)
    ctx.sync()
    s0 = arena.stats()
    t0 = time.perf_counter()
    arena.begin_move()
    arena.simulate(max(0, budget - steps_prof))
    ctx.sync()
    s1 = arena.stats()
    ctx.prof_enable(True, classes=[capi.PROF_SELECT, capi.PROF_EXPAND])
    arena.simulate(min(budget, steps_prof))
    ctx.sync()
    ctx.prof_enable(False)
    s2 = arena.stats()
    arena.end_move(True)
    ctx.sync()
    dt = time.perf_counter() - t0
    s3 = arena.stats()
    sel_n, sel_ms = ctx.prof_read(capi.PROF_SELECT)
    exp_n, exp_ms = ctx.prof_read(capi.PROF_EXPAND)

    def shape(a, b):
        sims = max(1, b["sims_total"] - a["sims_total"])
        path = (b["path_nodes"] - a["path_nodes"]) / sims
        levels = max(1, (b["path_nodes"] - a["path_nodes"]) - (b["sims_total"] - a["sims_total"]))
        return {"mean_path_nodes": path, "mean_children_per_select": (b["children_read"] - a["children_read"]) / levels}

    nodes = [arena.tree_nodes(g, a) for g in range(0, G, max(1, G // 32)) for a in (0, 1)]
    out = {"workload": "19x19, K=%d, %d blocks, %d games, %d sims/move, %s; policy FC scaled for a mean largest prior of %.2f%s"
                       % (K, L, G, budget, compute, target_max_prior, ", value output zeroed" if zero_value else ""),
           "net": info, "whole_move_seconds": dt, "sims_per_s": (s3["sims_nonnull"] - s0["sims_nonnull"]) / dt,
           "null_sims": (s3["sims_total"] - s0["sims_total"]) - (s3["sims_nonnull"] - s0["sims_nonnull"]),
           "tree_shape_whole_move": shape(s0, s2), "tree_shape_last_steps": shape(s1, s2),
           "k_select_ms_per_step": (sel_ms / sel_n) if sel_n else None, "k_expand_ms_per_step": (exp_ms / exp_n) if exp_n else None,
           "ms_per_step": dt / budget * 1e3,
           "mcts_share_of_step": ((sel_ms / sel_n + exp_ms / exp_n) / (dt / budget * 1e3)) if sel_n and exp_n else None,
           "max_path_nodes": arena.max_path_nodes(), "max_tree_nodes_sampled": int(max(nodes)), "tree_full": s3["tree_full"],
           "note": "k_select / k_expand: HIP events on the last %d steps of the move (the deepest trees); ms_per_step: the whole move / Budget" % steps_prof}
    arena.close()
    net.close()
    return out


def config0_leg():
    """BASELINE configs[0] exactly as the reference's README runs it — AZ.Learn(5, 50, 100, 100) on mnk.TicTacToe() with
    dual.DefaultConf(3, 3, 10), MCTS Budget 1000 — through the C++ host mirror over the C ABI (tests/cpp/az_learn_ttt, its own
    process): wall time incl. process start."""
    import subprocess
    exe = os.path.join(ROOT, "tests", "cpp", "az_learn_ttt")
    if not os.path.exists(exe):
        return {"error": "tests/cpp/az_learn_ttt not built"}
    t0 = time.perf_counter()
    out = subprocess.run([exe, "5", "50", "100", "100", "1000"], capture_output=True, text=True, timeout=120)
    wall = time.perf_counter() - t0
    lines = [l for l in out.stdout.splitlines() if l.startswith("epoch")]
    return {"workload": "configs[0]: AZ.Learn(5, 50, 100, 100), TicTacToe, DefaultConf(3,3,10), Budget 1000 (C++ host mirror over the C ABI)",
            "seconds": wall, "ok": "AZ_LEARN OK" in out.stdout, "epochs": lines}


def latency_leg(ctx, lanes_list=(1, 8, 16), moves=3, sims=1600):
    """BASELINE config #5 (tournament Agent.Search): 19x19, K=256, 40 blocks, 1600 sims/move, ONE tree through the single-tree
    boundary's engine; p50 wall time of a move (begin_move + simulate(Budget) + end_move + sync) on a short sample, at three operating
    points side by side: the sequential search (lanes 1, the declared semantics) and lane rounds of 8 and 16 — the
    deterministic restatement of what the reference itself does (runtime.NumCPU() goroutines on one tree with a stored virtual loss,
    mcts/search.go:112-131; bit-exact vs the oracle's parallelRound).  Towers: lanes 1 and 8 run the fp16x2 one-launch-per-layer kernel
    (AGZ_COMPUTE_AUTO, conv_lat.hpp: up to 8-11 boards), lanes 16 the Winograd fp16x2 tower kept at every batch size (AGZ_COMPUTE_FORCE).
    Each point carries the per-simulation kernel split (HIP events on ONE extra move, outside the timed ones) and the roofline of the
    tower layer: a 19x19 / K=256 dual layer reads its 4.72 MB fp16 hi/lo weight image first touch, 188.7 MB per 40-block evaluation."""
    S, K, L = 19, 256, 40
    net = A.Net(ctx, K, L, 2 * K, S, S, 18, S * S + 1, BatchSize=1, bn_mode=capi.BN_IDENTITY)
    net.init_random(1337)
    standard_bn_init(net)
    net.commit()
    w_layer = (K // 32) * 9 * 2 * (2 * K) * 32 * 2           # bytes of one layer's latency-regime weight image [c/32][tap][2][2K][32] fp16
    out = {"workload": "config #5: 19x19 wq Agent.Search, K=256, 40 blocks, %d sims/move, one tree" % sims, "moves_timed": moves,
           "tower_weight_bytes_per_eval": L * w_layer}
    for lanes in lanes_list:
        net.set_compute_mode((capi.COMPUTE_WINO_H2 | capi.COMPUTE_FORCE) if lanes > 8 else capi.COMPUTE_AUTO)   # AUTO up to 8 boards: the fp16x2 latency kernel (F32_MFMA keeps exact fp32 products)
        arena = A.Arena(ctx, capi.GAME_WQ, S, S, komi=7.5, encoder=capi.ENC_WQ, n_games=1, seed=7, Budget=sims)
        arena.set_inferencer(0, capi.INF_NET, net)
        arena.set_inferencer(1, capi.INF_NET, net)
        arena.set_parallel(lanes)
        arena.reset()
        arena.random_moves(np.array([60], np.int32), 1337)
        lat = []
        for mv in range(moves + 1):
            t0 = time.perf_counter()
            arena.begin_move()
            arena.simulate(sims)
            arena.end_move(False)
            ctx.sync()
            if mv >= 1:
                lat.append(time.perf_counter() - t0)
        # one extra move with every kernel class bracketed by HIP events: where a simulation's time goes (not part of the p50)
        ctx.prof_enable(True)
        arena.begin_move()
        arena.simulate(sims)
        arena.end_move(False)
        ctx.sync()
        ctx.prof_enable(False)
        split = {}
        for name, k in (("tower", capi.PROF_CONV), ("input_layer", capi.PROF_CONV_INIT), ("heads", capi.PROF_HEADS), ("select", capi.PROF_SELECT), ("expand", capi.PROF_EXPAND)):
            n, ms = ctx.prof_read(k)
            split[name + "_us_per_sim"] = ms * 1e3 / sims
        rounds = sims / lanes
        tower_ms_per_eval = split["tower_us_per_sim"] * 1e-3 * lanes       # one tower pass serves `lanes` simulations
        latency_tower = lanes <= 8
        out["lanes_%d" % lanes] = {"p50_move_s": float(np.percentile(lat, 50)), "max_move_s": float(np.max(lat)),
                                   "ms_per_sim": float(np.median(lat)) / sims * 1e3, "rounds_per_move": rounds,
                                   "kernel_split": split,
                                   "kernel_split_note": "HIP events around every launch on one extra move: each launch pays ~2-3 us of event overhead, so the classes sum to more than ms_per_sim (the timed moves run without events)",
                                   "tower": "fp16x2 one-launch-per-layer kernel (latency regime, conv_lat.hpp)" if latency_tower else "winograd fp16x2 (forced at every batch size)",
                                   "roofline": ({"bound": "hbm", "kernel": "conv3x3_lat_h2_kernel (one launch per dual layer)", "bytes_per_eval": L * w_layer,
                                                 "tower_ms_per_eval": tower_ms_per_eval, "us_per_layer": tower_ms_per_eval * 1e3 / L,
                                                 "achieved_GBs": L * w_layer / (tower_ms_per_eval * 1e-3) / 1e9, "peak_GBs": HBM_PEAK_GBS,
                                                 "frac": L * w_layer / (tower_ms_per_eval * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                 "note": "first-touch weight stream of a batch-%d evaluation; the layer is bound by its launch boundary + first-byte latency + 64 B/clk "
                                                         "LDS fill per CU, not by HBM (DESIGN 4c)" % lanes} if latency_tower else None)}
        arena.close()
    net.close()
    return out


def train_leg(ctx, steps=10):
    """SURVEY 8(f)-1 under the driver's clock: one dual.Train batch (dualnet/meta.go:16-54) of the config #4 network —
    19x19, K=256, 20 blocks, BatchSize 256 — forward (training-mode BatchNorm), loss, backward, vanilla SGD step, in the trainer's
    AGZ_COMPUTE_WINO_H2 arithmetic (every gradient tensor within 2e-5 * max|g| of the oracle: tests/test_train_gpu.py); host batch in,
    cost out, `steps` timed steps after one warm-up."""
    S, K, L, B = 19, 256, 20, 256
    t = A.Trainer(ctx, K, L, 2 * K, S, S, 18, S * S + 1, B)
    t.init_random(1337)
    t.set_compute_mode(capi.COMPUTE_WINO_H2)
    rng = np.random.default_rng(0)
    x = rng.choice(np.array([-1, 0, 1], np.float32), size=(B, 18, S, S)).astype(np.float32)
    pi = np.zeros((B, S * S + 1), np.float32)
    pi[np.arange(B), rng.integers(0, S * S + 1, B)] = 1
    v = rng.choice(np.array([-1, 0, 1], np.float32), size=B).astype(np.float32)
    c = t.batch(x, pi, v)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        c = t.batch(x, pi, v)
    ctx.sync()
    dt = (time.perf_counter() - t0) / steps
    # the same call with the batch in page-locked host memory (agz_host_alloc: what the Go shim's staging uses)
    xp, pp, vp = ctx.host_array(x.shape), ctx.host_array(pi.shape), ctx.host_array(v.shape)
    xp[...], pp[...], vp[...] = x, pi, v
    t.batch(xp, pp, vp)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        t.batch(xp, pp, vp)
    ctx.sync()
    dt_pinned = (time.perf_counter() - t0) / steps
    for a_ in (xp, pp, vp):
        ctx.host_free(a_)
    hw = S * S
    flops = 3 * (2.0 * 18 * K * 9 * hw + L * 2 * 2.0 * K * K * 9 * hw) * B   # forward + data gradient + weight gradient, direct-equivalent
    # the loop AZ.Learn runs (agogo.go:123-133 -> dual.Train, dualnet/meta.go:16-54) with the examples RESIDENT in HBM (agz_examples_* ->
    # agz_train_dev): `steps` batches of one iteration, gathered device to device by the shuffled row index, no per-batch host round trip —
    # the per-batch cost of dual.Train as the path runs it (the figure above pays a 27 MB pageable copy and a synchronisation per batch)
    dev_ms = None
    try:
        ex = A.Examples(ctx, 18, S, S, S * S + 1)
        for _ in range(steps):
            ex.append_host(x, pi, v)
        nb = ex.prepare(B, 0, seed=1)
        xd, pd, vd, rows_d, bd = ex.tensors_dev()
        t.train_dev(xd, pd, vd, 1, 1, seed=3)           # warm
        ctx.sync()
        t0 = time.perf_counter()
        t.train_dev(xd, pd, vd, bd, 1, seed=3)
        ctx.sync()
        dev_ms = (time.perf_counter() - t0) / bd * 1e3
        ex.close()
    except Exception as e:   # the leg above stands on its own
        dev_ms = {"error": repr(e)}
    t.close()
    return {"workload": "dual.Train batch: 19x19, K=256, 20 blocks, BatchSize 256 (config #4 network), trainer AGZ_COMPUTE_WINO_H2",
            "steps_timed": steps, "step_ms": dt * 1e3, "examples_per_s": B / dt, "direct_equivalent_tflops": flops / dt / 1e12,
            "cost": float(c), "note": "includes the host -> device copy of the batch (27 MB) and the cost read-back: the boundary's agz_trainer_batch",
            "step_ms_pinned_host_batch": dt_pinned * 1e3,
            "step_ms_device_resident": dev_ms,
            "step_ms_device_resident_note": "agz_train_dev over %d batches of examples resident in HBM (what AZ.Learn's dual.Train runs on): per batch, no host copy, "
                                            "no per-batch synchronisation" % steps}


def launcher_command(argv, gpus, environ, port=None):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: the command that re-executes this script as N ranks,
    one per GPU (the contract's own launch line: torch.distributed.run, 127.0.0.1 rendezvous).  None when there is nothing to
    re-launch (N = 1, or RANK / WORLD_SIZE already set by a launcher)."""
    if gpus <= 1 or ("RANK" in environ and "WORLD_SIZE" in environ):
        return None
    if port is None:
        import socket
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
        s_.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def aggregate_over_ranks(dt, sims, evals, iters, rank, world, device=None):
    """The bench line's aggregation: MAX over ranks of the timed region's wall time, SUM of the counters; every rank's own
    non-null simulations travel along (per_rank_sims) so that the line is self-checking.  Runs on any process group (RCCL on the
    GPUs, gloo in tests/test_dist_gloo.py)."""
    per_rank = [0.0] * world
    per_rank[rank] = float(sims)
    t_max, sums = adist.reduce_step_timing(dt, [sims, evals, iters] + per_rank, device=device)
    return {"t_max": t_max, "sims": sums[0], "evals": sums[1], "iters": sums[2], "per_rank_sims": sums[3:], "world_size": world}


def headline_fields(agg, gpus, steps, warmup):
    """metric / value / n_gpus / ms_per_step of the line from the aggregate; refuses to print a line whose n_gpus is not the
    --gpus it was asked for, or whose per-rank counters do not add up to the aggregate."""
    if agg["world_size"] != gpus:
        raise SystemExit("bench.py: --gpus %d but %d rank(s) took part" % (gpus, agg["world_size"]))
    if len(agg["per_rank_sims"]) != gpus or abs(sum(agg["per_rank_sims"]) - agg["sims"]) > 0.5 or min(agg["per_rank_sims"]) <= 0:
        raise SystemExit("bench.py: per-rank simulations %r do not add up to %r on %d rank(s)" % (agg["per_rank_sims"], agg["sims"], gpus))
    return {"metric": "mcts_sims_per_sec", "value": agg["sims"] / agg["t_max"], "unit": "sims/s", "n_gpus": gpus,
            "steps": steps, "warmup": warmup, "ms_per_step": agg["t_max"] / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--games", type=int, default=512, help="concurrent self-play games per GPU")
    ap.add_argument("--size", type=int, default=19)
    ap.add_argument("--K", type=int, default=256)
    ap.add_argument("--L", type=int, default=20)
    ap.add_argument("--budget", type=int, default=800, help="simulations per move")
    ap.add_argument("--max-nodes", type=int, default=0, help="node pool per tree (0: the library default, four searches' worth of expansions)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=64, help="threads of the cpu_baseline legs (0 = every host core: slower in aggregate and ~7 minutes)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=30.0, help="size of the cpu_baseline's 19x19 leg (seconds of wall time it is sized for)")
    ap.add_argument("--no-games-leg", action="store_true", help="skip the measured games/s leg (config #2)")
    ap.add_argument("--two-nets", action="store_true", help="agents A and B hold different networks")
    ap.add_argument("--compute", choices=["wino_h2", "wino", "bf16x3", "f32", "fp16x2"], default="wino_h2",
                    help="dual-block conv arithmetic: wino_h2 = Winograd F(5x5,3x3) (F(4x4,3x3) where that needs fewer rows), fp32 transforms, the transform-domain operand "
                         "written pre-split into two fp16 pieces (per-board power-of-two range from a proven bound), 3 fp16 MFMAs per "
                         "product; wino = the same with bf16x3 products (6 MFMAs per product); bf16x3 = direct conv, exact "
                         "3-way bf16 split on the bf16 matrix pipe (fp32-grade), f32 = v_mfma_f32_32x32x2_f32, fp16x2 = "
                         "range-managed 2-way fp16 split (3 MFMAs per product; opt-in fast mode)")
    ap.add_argument("--no-f32-leg", action="store_true", help="skip the short comparison legs in the other compute modes")
    ap.add_argument("--empty-boards", action="store_true", help="round-1 workload: all games start from the empty board")
    ap.add_argument("--no-pregrow", action="store_true",
                    help="skip the untimed whole move and the tree pre-growth (the timed steps then run on the first simulations of a move)")
    ap.add_argument("--no-latency-leg", action="store_true", help="skip the configs[4] single-tree move-latency sample")
    ap.add_argument("--no-go9-leg", action="store_true", help="skip the measured 9x9 games/s leg (configs[2])")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the dual.Train step leg (SURVEY 8(f)-1)")
    ap.add_argument("--no-deep-leg", action="store_true", help="skip the deep-tree legs (the headline workload under a peaked policy head)")
    ap.add_argument("--no-complete-games-leg", action="store_true", help="skip the 512 complete 19x19 games at a small Budget (measured game length)")
    ap.add_argument("--complete-games-budget", type=int, default=4,
                    help="simulations per move of the complete-games leg (4: ~40 s; the 16-simulation run of the same leg — 168 s — is committed as "
                         "profiles/r06/complete_games_19x19_budget16.json: mean length 692 of a 722-move cap, 473 of 512 games end at the cap)")
    ap.add_argument("--prof-stride", type=int, default=4,
                    help="inside the timed region every N-th launch of the dominant kernel is bracketed with HIP events (0: none)")
    ap.add_argument("--tower-queues", type=int, default=0, choices=[0, 1, 2],
                    help="agz_net_set_tower_queues: 0 = library default (two queues from 256 boards), 1, 2.  With two queues the per-kernel "
                         "durations of the roofline are taken on isolated one-queue steps right after the timed region")
    ap.add_argument("--shared-gpu", action="store_true",
                    help="debug: all ranks use GPU 0 and gloo collectives (exercises the N>1 code path on a 1-GPU box)")
    ap.add_argument("--print-launch", action="store_true", help="print the re-launch command for --gpus N (JSON list) and exit")
    args = ap.parse_args()
    t_process = time.perf_counter()

    # `python bench.py --gpus N` (N > 1) outside a launcher: become the N-rank job (one process per GPU over RCCL)
    cmd = launcher_command([a_ for a_ in sys.argv[1:] if a_ != "--print-launch"], args.gpus, os.environ)
    if args.print_launch:
        print(json.dumps(cmd))
        return
    if cmd is not None:
        sys.stdout.flush()
        os.execv(cmd[0], cmd)

    # the contract is ONE JSON line on stdout: libraries (gloo's "[Gloo] Rank ..." lines, RCCL's version banner) write to fd 1 too,
    # so fd 1 is pointed at stderr for the whole run and the line goes to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # (rank 0 times the CPU baseline — about a minute — while the other ranks wait in the closing barrier: a generous group time-out)
    rank, local, world = adist.init_from_env(backend="gloo" if args.shared_gpu else None, timeout_s=3600)
    if args.shared_gpu:
        local = 0
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE %d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libagz has no CPU fallback)")
    torch.cuda.set_device(local)
    ctx = A.Ctx(local)
    S, K, L, G = args.size, args.K, args.L, args.games
    Aspace = S * S + 1
    nets = []
    for i in range(2 if args.two_nets else 1):
        net = A.Net(ctx, K, L, 2 * K, S, S, 18, Aspace, bn_mode=capi.BN_IDENTITY)
        net.init_random(1337 + i)
        standard_bn_init(net)
        net.commit()
        net.set_compute_mode(MODES[args.compute])
        net.set_tower_queues(args.tower_queues)
        nets.append(net)
    arena = A.Arena(ctx, capi.GAME_WQ, S, S, komi=7.5, encoder=capi.ENC_WQ, n_games=G, seed=1337 + rank, max_nodes=args.max_nodes,
                    Budget=args.budget, PUCT=1.0, RandomCount=0, DumbPass=True,
                    PassPreference=capi.DONT_PREFER_PASS)
    arena.set_inferencer(0, capi.INF_NET, nets[0])
    arena.set_inferencer(1, capi.INF_NET, nets[-1])
    arena.reset()
    # every game on its own position: u ~ U[0, floor(0.6*H*W)] random legal moves, deterministic per (rank, slot)
    open_rng = np.random.default_rng(1337 + rank)
    n_open = open_rng.integers(0, int(0.6 * S * S) + 1, size=G).astype(np.int32)
    if args.empty_boards:
        n_open[:] = 0
    arena.random_moves(n_open, 1337 + 7919 * rank)

    sims_in_move = [0]

    def step():
        if sims_in_move[0] == 0:
            arena.begin_move()
        arena.simulate(1)
        sims_in_move[0] += 1
        if sims_in_move[0] == args.budget:
            arena.end_move(True)
            sims_in_move[0] = 0

    def fence():
        ctx.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # (1) one whole move, untimed for `value` but measured: begin_move + Budget simulations + end_move for all games
    full_move = None
    if not args.no_pregrow:
        fence()
        m0 = arena.stats()
        g0 = time.perf_counter()
        for _ in range(args.budget):
            step()
        fence()
        d_move = time.perf_counter() - g0
        m1 = arena.stats()
        full_move = {"seconds": d_move, "moves": m1["moves_played"] - m0["moves_played"],
                     "moves_per_s": (m1["moves_played"] - m0["moves_played"]) / d_move,
                     "sims_per_s": (m1["sims_nonnull"] - m0["sims_nonnull"]) / d_move,
                     "note": "one whole move of all %d games from their random openings: begin_move (re-root/prepareRoot) + %d "
                             "simulations + end_move (bestMove, example, Apply, Ended); measured, not derived" % (G, args.budget)}
        # (2) grow the next move's trees so that the timed steps straddle the following move boundary
        pre = max(0, args.budget - args.warmup - args.steps // 2)
        for _ in range(pre):
            step()
    for _ in range(args.warmup):
        step()
    fence()
    st0 = arena.stats()
    hbm_free, hbm_total = torch.cuda.mem_get_info(local)   # device-wide: on a shared GPU (--shared-gpu) every rank's allocations count
    # inside the timed region only the dominant kernel class and the move-boundary class record HIP events (two event records
    # per launch; timing all nine classes cost ~1.3 ms of a 24 ms step in round 2's first run); the full breakdown is taken
    # on extra steps after the timed region
    dom = capi.PROF_WINO_GEMM if args.compute in ("wino", "wino_h2") else capi.PROF_CONV
    # ... and of the dominant kernel only every --prof-stride-th launch (default 4: 100+ of the timed region's 400+ launches):
    # the roofline needs the kernel's AVERAGE duration over the region, and each bracketed launch costs two event records
    # Two queues (the library default from 256 boards for wino_h2): the half-batch chains overlap, every kernel's own duration
    # inflates and only queue 0's launches could be bracketed — per-kernel time is then taken on the isolated one-queue steps below
    two_queues = args.compute == "wino_h2" and G >= 256 and args.tower_queues != 1
    in_region_prof = args.prof_stride > 0 and not two_queues
    if in_region_prof:
        ctx.prof_set_stride(dom, args.prof_stride)
        ctx.prof_enable(True, classes=[dom, capi.PROF_MOVE])
    elif two_queues:
        ctx.prof_enable(True, classes=[capi.PROF_MOVE])
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    ctx.prof_enable(False)
    st1 = arena.stats()
    dom_n, dom_ms = ctx.prof_read(dom)
    move_n, _ = ctx.prof_read(capi.PROF_MOVE)
    ctx.prof_set_stride(dom, 1)
    if two_queues:
        for n_ in nets:
            n_.set_tower_queues(1)
    for _ in range(2):      # (settle on the one-queue plan before the bracketed steps)
        step()
    fence()
    ctx.prof_enable(True)
    for _ in range(N_ONE_QUEUE_STEPS):
        step()
    fence()
    ctx.prof_enable(False)
    if two_queues:
        for n_ in nets:
            n_.set_tower_queues(args.tower_queues)

    max_path_main = arena.max_path_nodes()
    sims = st1["sims_nonnull"] - st0["sims_nonnull"]
    sims_all = st1["sims_total"] - st0["sims_total"]
    evals = st1["nn_evals"] - st0["nn_evals"]
    # MAX of the wall time, SUM of the counters over ranks; the per-rank sims travel along so that the line is self-checking
    agg = aggregate_over_ranks(dt, sims, evals, sims_all, rank, world, device="cuda" if world > 1 else None)
    t_max, sims_sum, evals_sum, iters_sum, per_rank_sims = agg["t_max"], agg["sims"], agg["evals"], agg["iters"], agg["per_rank_sims"]
    prof = {}
    for name, k in (("conv_dual", capi.PROF_CONV), ("conv_init", capi.PROF_CONV_INIT), ("heads", capi.PROF_HEADS),
                    ("select", capi.PROF_SELECT), ("expand", capi.PROF_EXPAND), ("move", capi.PROF_MOVE),
                    ("wino_in", capi.PROF_WINO_IN), ("wino_gemm", capi.PROF_WINO_GEMM), ("wino_out", capi.PROF_WINO_OUT)):
        n, ms = ctx.prof_read(k)
        prof[name] = {"launches": n, "avg_ms": (ms / n) if n else None, "total_ms": ms}
    dom_name = "wino_gemm" if args.compute in ("wino", "wino_h2") else "conv_dual"
    prof[dom_name + "_timed_region"] = {"launches": dom_n, "avg_ms": (dom_ms / dom_n) if dom_n else None, "total_ms": dom_ms}
    prof["breakdown_note"] = ("all classes: %d extra steps after the timed region%s; *_timed_region: HIP events inside the timed region"
                              % (N_ONE_QUEUE_STEPS, " on ONE queue (the timed region runs the tower on two)" if two_queues else ""))
    if not in_region_prof:
        prof[dom_name + "_timed_region"] = {"launches": 0, "avg_ms": None, "total_ms": 0.0}

    # short comparison legs in the other compute modes (same arena, the games simply continue)
    legs = {}
    if world == 1 and not args.no_f32_leg:
        for mode in ("f32", "bf16x3", "fp16x2", "wino", "wino_h2"):
            if mode == args.compute:
                continue
            for n_ in nets:
                n_.set_compute_mode(MODES[mode])
            k2 = max(2, min(args.steps, 8))
            step(); fence()
            s0 = arena.stats()
            ctx.prof_enable(True)
            fence()
            g0 = time.perf_counter()
            for _ in range(k2):
                step()
            fence()
            d2 = time.perf_counter() - g0
            ctx.prof_enable(False)
            s1 = arena.stats()
            n2, ms2 = ctx.prof_read(capi.PROF_CONV)
            legs[mode] = {"steps": k2, "sims_per_s": (s1["sims_nonnull"] - s0["sims_nonnull"]) / d2, "ms_per_step": d2 / k2 * 1e3,
                          "conv_dual_avg_ms": (ms2 / n2) if n2 else None}
        for n_ in nets:
            n_.set_compute_mode(MODES[args.compute])

    # the one exchange step of the path, untimed for `value`: the examples recorded so far gathered over all ranks —
    # inside libagz (agz_comm_* / agz_examples_allgather: RCCL grouped broadcasts over xGMI), the way the Go host would do it
    gather = None
    gather_hung = False
    if world > 1:
        def gather_leg():
            try:
                ex = A.Examples(ctx, 18, S, S, Aspace)
                comm = adist.make_comm(ctx)
                # (agz_examples_append_arena takes finished games only; the bench's games are mid-way, so the arena rows go in raw)
                import ctypes as C
                pp, po, pv, n = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_int32(0)
                capi._check(capi.lib().agz_arena_examples_dev(arena.h, C.byref(pp), C.byref(po), C.byref(pv), C.byref(n)), "examples_dev")
                if n.value:
                    ex.append_dev(pp.value, po.value, pv.value, n.value)
                fence()
                g0 = time.perf_counter()
                comm.allgather_examples(ex)
                ctx.sync()
                g_ms = (time.perf_counter() - g0) * 1e3
                res = {"path": "libagz: agz_comm_init_rank + agz_examples_allgather (RCCL, one grouped set of broadcasts)",
                       "note": "self-check: rccl_ranks must equal n_gpus and rows_gathered the sum of every rank's rows_this_rank",
                       "ms": g_ms, "rows_this_rank": n.value, "rows_gathered": len(ex), "rccl_ranks": comm.size(), "rccl_rank": comm.rank(),
                       "GB_per_s": len(ex) * (18 * S * S + Aspace + 1) * 4 / (g_ms * 1e-3) / 1e9}
                comm.close()
                ex.close()
                return res
            except Exception as e:   # never lose the bench line to the untimed leg
                return {"error": repr(e)}
        # ... nor to a collective that never returns (N > 1 over xGMI has not been run anywhere yet): the leg runs on its own thread
        # (ctypes calls release the GIL) and is given two minutes; a hung leg is reported as such and the process then leaves
        # through os._exit after the line is out
        import threading
        box = {}
        th = threading.Thread(target=lambda: box.__setitem__("gather", gather_leg()), daemon=True)
        th.start()
        th.join(120.0)
        if th.is_alive():
            gather, gather_hung = {"error": "the example all-gather leg did not return within 120 s"}, True
        else:
            gather = box.get("gather")
    # every rank's own row count, over the process group (not RCCL-in-libagz: an independent path), so that the line can check
    # rows_gathered = the sum; same two-minute rule (a rank stuck in the leg above never joins this reduction)
    rows_per_rank = None
    if world > 1 and not gather_hung:
        per = [0.0] * world
        per[rank] = float((gather or {}).get("rows_this_rank", -1))
        box2 = {}
        th2 = threading.Thread(target=lambda: box2.__setitem__("rows", adist.reduce_step_timing(0.0, per, device="cuda")[1]), daemon=True)
        th2.start()
        th2.join(120.0)
        if th2.is_alive():
            gather_hung = True
        else:
            rows_per_rank = [int(x) for x in box2.get("rows", [])]
    gather_ms = gather
    # device memory in use on every rank's GPU when the timed region started (all games, trees and network scratch allocated)
    hbm_per_rank = None
    if world > 1 and not gather_hung:
        per = [0.0] * world
        per[rank] = float(hbm_total - hbm_free)
        hbm_per_rank = [int(x) for x in adist.reduce_step_timing(0.0, per, device="cuda")[1]]
    else:
        hbm_per_rank = [int(hbm_total - hbm_free)]

    if rank == 0:
        # the search kernels of one step: algorithmic bytes from the live tree statistics of the timed region (SURVEY 8(d):
        # Select reads path nodes x children x 12 B — prior, visits, blackScores; expansion writes n_legal x 20 B of new nodes and the
        # leaf's encoded planes), duration from the HIP-event steps after the timed region; HBM counters QUOTED from the committed pass
        path_per_sim = (st1["path_nodes"] - st0["path_nodes"]) / max(1, sims_all)
        kids_per_node = (st1["children_read"] - st0["children_read"]) / max(1, (st1["path_nodes"] - st0["path_nodes"]) - sims_all)
        sel_ms, exp_ms = prof["select"]["avg_ms"], prof["expand"]["avg_ms"]
        sel_bytes = G * ((path_per_sim - 1) * kids_per_node * 12.0 + 18 * S * S * 4.0)      # child blocks read + the leaf's 18 planes written
        exp_bytes = G * ((S * S + 1) * 20.0 + (S * S + 1) * 4.0)                                # <= A new nodes + the policy row read
        pmc_mcts = None
        try:
            pmc_mcts = {k: {"hbm_bytes_per_launch": v["hbm_bytes_per_launch"], "GBps": v["GBps"], "frac_of_8TBps": v["frac_of_8TBps"]}
                        for k, v in json.load(open(os.path.join(ROOT, "profiles", "r05", "pmc_mcts_kernels.json")))["kernels"].items()}
        except Exception:
            pass
        mcts_detail = {"mean_path_nodes": path_per_sim, "mean_children_per_select": kids_per_node, "max_path_nodes": max_path_main,
                       "k_select": {"avg_ms": sel_ms, "algorithmic_bytes": sel_bytes, "GBps": (sel_bytes / (sel_ms * 1e-3) / 1e9) if sel_ms else None,
                                    "frac_of_8TBps": (sel_bytes / (sel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if sel_ms else None},
                       "k_expand": {"avg_ms": exp_ms, "algorithmic_bytes": exp_bytes, "GBps": (exp_bytes / (exp_ms * 1e-3) / 1e9) if exp_ms else None,
                                    "frac_of_8TBps": (exp_bytes / (exp_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if exp_ms else None},
                       "counters": pmc_mcts, "counters_source": "profiles/r05/pmc_mcts_kernels.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload; quoted)",
                       "note": "one 64-lane wavefront per game walks its own tree with the board in LDS: latency-bound by construction, 1-2 % of the HBM peak, "
                               "~1.3 % of the step; measured on near-uniform priors (BatchNorm gamma = 1 / beta = 0, identity statistics: config.weights) — trees are wide "
                               "(~250 children per visited node, paths ~3 nodes)"}
        flops_eval = nets[0].flops_per_eval()
        hw = S * S
        conv_flops_launch = 2.0 * (G * hw) * (2 * K) * (9 * K)  # algorithmic FLOPs of one dual-block launch
        conv_ms = prof["conv_dual"]["avg_ms"]
        achieved = conv_flops_launch / (conv_ms * 1e-3) / 1e12 if conv_ms else None
        peaks = {"f32": FP32_MFMA_PEAK_TFLOPS, "bf16x3": BF16X3_PEAK_TFLOPS, "fp16x2": FP16X2_PEAK_TFLOPS, "wino": BF16X3_PEAK_TFLOPS,
                 "wino_h2": FP16X2_PEAK_TFLOPS}
        peak = peaks[args.compute]
        wino_detail = None
        wtm = capi.wino_h2_tile(S, S) if args.compute == "wino_h2" else 4   # Winograd tile size of the mode (F(m x m, 3x3))
        npos = (wtm + 2) ** 2
        flops_launch, launch_ms, n_launch = conv_flops_launch, conv_ms, prof["conv_dual"]["launches"]
        if dom_name == "conv_dual" and prof["conv_dual_timed_region"]["avg_ms"]:   # direct modes: the block kernel itself, timed in the region
            launch_ms, n_launch = prof["conv_dual_timed_region"]["avg_ms"], prof["conv_dual_timed_region"]["launches"]
            achieved = flops_launch / (launch_ms * 1e-3) / 1e12
        if args.compute in ("wino", "wino_h2") and prof["wino_gemm"]["avg_ms"]:
            # dominant kernel of this mode: the transform-domain GEMMs of one block (its own FLOPs, not the direct conv's).
            # F(m x m, 3x3): npos = (m + 2)^2 positions, ceil(S/m)^2 tiles per board; wino = m 4, wino_h2 = agz_wino_h2_tile (5 on 19x19)
            tiles = G * ((S + wtm - 1) // wtm) ** 2
            flops_launch = 2.0 * npos * tiles * K * (2 * K)
            launch_ms, n_launch = prof["wino_gemm_timed_region"]["avg_ms"], prof["wino_gemm_timed_region"]["launches"]
            if not launch_ms:   # --prof-stride 0: nothing bracketed inside the timed region -> the breakdown steps' average
                launch_ms, n_launch = prof["wino_gemm"]["avg_ms"], prof["wino_gemm"]["launches"]
            achieved = flops_launch / (launch_ms * 1e-3) / 1e12
            in_bytes = 4.0 * (G * hw * K + npos * tiles * K)            # x read once + V written
            w_bytes = float(npos) * K * (2 * K) * (4 if args.compute == "wino_h2" else 6)   # the block's Winograd-domain weight image, read once
            gemm_bytes = 4.0 * (npos * tiles * K + npos * tiles * 2 * K) + w_bytes  # V read once + M written once + weights
            out_bytes = 4.0 * (npos * tiles * 2 * K + G * hw * K)       # M read + y written
            # the chained block (conv_wino_h2c.hpp): the kernel counted under wino_out reads M(l) and writes V2(l+1) — y never leaves the
            # chip between blocks; the input transform runs once per tower (block 0), the last block's output kernel writes y
            chained = args.compute == "wino_h2" and capi.wino_h2_chained(S, S, K) == 1
            if chained:
                out_bytes = 4.0 * (npos * tiles * 2 * K + npos * tiles * K)
            block_bytes = (gemm_bytes + out_bytes + (in_bytes + 4.0 * G * hw * K) / L) if chained else (in_bytes + gemm_bytes + out_bytes)
            def gbs(b, ms):
                return (b / (ms * 1e-3) / 1e9) if ms else None
            wino_detail = {
                "block_avg_ms": conv_ms, "block_direct_equivalent_tflops": conv_flops_launch / (conv_ms * 1e-3) / 1e12 if conv_ms else None,
                "tile": wtm, "positions": npos, "tiles_per_launch": tiles,
                "block_note": "one dual block = %d GEMMs + (chained) the fused output-transform / epilogue / next block's input-transform kernel "
                              "[three-kernel form: input transform + GEMMs + output transform]; direct-equivalent = the "
                              "FLOPs a direct 3x3 convolution would need for the same result (%.2fx the GEMM FLOPs here)"
                              % (npos, conv_flops_launch / flops_launch),
                "wino_in": {"avg_ms": prof["wino_in"]["avg_ms"], "bound": "hbm", "algorithmic_bytes": in_bytes,
                            "achieved_GBs": gbs(in_bytes, prof["wino_in"]["avg_ms"]), "peak_GBs": HBM_PEAK_GBS},
                "wino_gemm": {"avg_ms": launch_ms, "bound": "hbm" if args.compute == "wino_h2" else "mfma", "flops": flops_launch,
                              "algorithmic_bytes": gemm_bytes, "achieved_GBs": gbs(gemm_bytes, launch_ms), "peak_GBs": HBM_PEAK_GBS,
                              "tflops": achieved, "mfma_frac": achieved / peak, "mfma_peak_tflops": peak},
                "chained": chained,
                "block_algorithmic_bytes": block_bytes,
                "block_achieved_GBs": gbs(block_bytes, conv_ms),
                "wino_out": {"avg_ms": prof["wino_out"]["avg_ms"], "bound": "hbm", "algorithmic_bytes": out_bytes,
                             "achieved_GBs": gbs(out_bytes, prof["wino_out"]["avg_ms"]), "peak_GBs": HBM_PEAK_GBS}}
        traffic = None
        pmc_name = {"f32": "pmc_conv_dual.json", "bf16x3": "pmc_conv_x3.json", "fp16x2": "pmc_conv_h2.json",
                    "wino": "pmc_wino_gemm.json", "wino_h2": "pmc_wino_h2_gemm.json"}[args.compute]
        if args.compute == "wino_h2" and capi.wino_h2_chained(S, S, K) == 1:
            pmc_name = "pmc_wino_h2c.json"   # the chained block's GEMM (wino_gemm_h2g_kernel), round 4
        pmc_path = os.path.join(ROOT, "profiles", pmc_name)
        for rd in ("r05", "r06"):   # the newest pass of the same kernel (scripts/r5_pmc_tower.sh, re-run by scripts/r6_evidence.sh)
            if os.path.exists(os.path.join(ROOT, "profiles", rd, pmc_name)):
                pmc_path = os.path.join(ROOT, "profiles", rd, pmc_name)
        if os.path.exists(pmc_path) and (S, K, L, G) == (19, 256, 20, 512):  # the PMC pass was taken on this exact shape
            try:
                traffic = json.load(open(pmc_path)).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        traffic_source = os.path.relpath(pmc_path, ROOT) if traffic is not None else None
        # the memory system's own ceiling for this kernel's byte mix, MEASURED (scripts/probes/rw_probe.hip) and committed — not a constant
        stream_ceiling, stream_src = None, None
        try:
            rw = json.load(open(os.path.join(ROOT, "profiles", "r05", "rw_probe.json")))
            stream_ceiling, stream_src = [float(x) * 1e3 for x in rw["gemm_mix_1_2_TBps"]], "profiles/r05/rw_probe.json (gemm_mix_1_2_TBps: every box and grid measured)"
        except Exception:
            pass
        for mode, leg in legs.items():
            if leg.get("conv_dual_avg_ms"):
                leg["conv_dual_tflops"] = conv_flops_launch / (leg["conv_dual_avg_ms"] * 1e-3) / 1e12
                if mode in ("wino", "wino_h2"):
                    leg["note"] = "direct-equivalent FLOPs per block time (the Winograd modes execute 3.6-4.1x fewer)"
                else:
                    leg["frac_of_its_roofline"] = leg["conv_dual_tflops"] / peaks[mode]
                    leg["roofline_peak_tflops"] = peaks[mode]
        dtypes = {"f32": "f32",
                  "bf16x3": "f32 (bf16x3 split: each fp32 operand = 3 exact bf16 pieces, 6 bf16 MFMAs per product, fp32 accumulate)",
                  "fp16x2": "f32 (fp16x2 split: power-of-two range scaling, 2 fp16 pieces = 23 significand bits, 3 fp16 MFMAs per product, fp32 accumulate)",
                  "wino": "f32 (Winograd F(4x4,3x3): fp32 transforms; transform-domain products as bf16x3 = 3 exact bf16 pieces per fp32 operand, 6 bf16 MFMAs per product, fp32 accumulate)",
                  "wino_h2": "f32 (Winograd F(%dx%d,3x3): fp32 transforms; transform-domain products as fp16x2 = each fp32 operand scaled by a power of two and split into 2 fp16 pieces (22-23 significand bits, absolute error <= 2^-38 of the tensor range), 3 fp16 MFMAs per product, fp32 accumulate)" % (wtm, wtm)}
        kernels = {"f32": "conv3x3_mfma_kernel<2,2,2,DUAL> (fused dual-branch block)",
                   "bf16x3": "conv3x3_x3_kernel<DUAL> (fused dual-branch block, bf16x3)",
                   "fp16x2": "conv3x3_h2w_kernel (fused dual-branch block, fp16x2, 128x256 tile)",
                   "wino": "wino_gemm_kernel (36 transform-domain GEMMs of one dual block, bf16x3 products; 0.70 of the block's 1.09 ms)",
                   "wino_h2": "wino_gemm_h2g_kernel<8> (the %d transform-domain GEMMs of one dual block, F(%dx%d,3x3), fp16x2 products, 128x256 tile, operands DMA'd into LDS (buffer_load ... lds), three workgroups per CU; the larger of the chained block's two kernels)" % (npos, wtm, wtm)}
        notes = {"f32": "dense fp32 MFMA peak",
                 "bf16x3": ("algorithmic fp32-grade FLOPs against the dense bf16 MFMA peak / 6 (six bf16 MFMAs per product); the same "
                            "FLOPs are %.2fx the fp32-MFMA peak of 157.3; measured bare-MFMA ceiling under the power cap on random "
                            "rotating operands: 1772 TFLOP/s bf16 = 295 in these units (DESIGN.md 4b)" % ((achieved or 0) / FP32_MFMA_PEAK_TFLOPS)),
                 "fp16x2": ("algorithmic FLOPs against the dense fp16 MFMA peak / 3 (three fp16 MFMAs per product); %.2fx the "
                            "fp32-MFMA peak (DESIGN.md 4c)" % ((achieved or 0) / FP32_MFMA_PEAK_TFLOPS)),
                 "wino": ("FLOPs of the transform-domain GEMMs (what this formulation executes) against the dense bf16 MFMA peak / 6; "
                          "the block as a whole delivers the direct convolution's result at extra.wino.block_direct_equivalent_tflops "
                          "(DESIGN.md 4d); the two transform kernels are HBM-bound, see extra.wino"),
                 "wino_h2": ("with 3 fp16 MFMAs per product the GEMMs' arithmetic intensity (3 x %.1f GFLOP of MFMA work over %.2f GB = %.0f FLOP/B) "
                             "sits BELOW the ridge (2516.6 TFLOP/s / 8 TB/s = 315 FLOP/B): the kernel's roofline is the HBM roof. achieved = "
                             "algorithmic bytes (V read once + M written once + the block's weights) / launch time; the MFMA-side fraction "
                             "(transform-domain FLOPs against bf16 peak / 3) is extra.wino.wino_gemm.mfma_frac (DESIGN.md 4e)"
                             % ((flops_launch / 1e9, wino_detail["wino_gemm"]["algorithmic_bytes"] / 1e9,
                                 3 * flops_launch / wino_detail["wino_gemm"]["algorithmic_bytes"]) if wino_detail else (0, 0, 0)))}
        out = headline_fields(agg, args.gpus, args.steps, args.warmup) | {
            "dtype": dtypes[args.compute],
            "data": "synthetic",
            "config": {"workload": "19x19 Go (wq) self-play: K=%d, %d dual-branch blocks, FC=%d, A=%d, WQEncoder F=18, "
                                   "%d concurrent games/GPU, %d sims/move, leaf batch=%d, one net for both agents=%s"
                                   % (K, L, 2 * K, Aspace, G, args.budget, G, str(not args.two_nets)),
                       "board": S, "K": K, "blocks": L, "games_per_gpu": G, "sims_per_move": args.budget,
                       "weights": "random-init seed 1337: GlorotU conv, GlorotN FC as the reference (ermahagerdmonards.go:39,80); DEVIATION from the "
                                  "reference initialiser: BatchNorm gamma=1, beta=0 with IDENTITY statistics instead of GlorotN gamma/beta under "
                                  "the degenerate-eps reading (x316 per layer saturates softmax/tanh and every search tree degenerates); "
                                  "FLOPs and bytes per evaluation are identical",
                       "priors_note": "with gamma = 1 / beta = 0 and identity statistics the random-init policy is near uniform: the search runs on WIDE trees "
                                      "(~250 children per visited node, paths ~3 nodes: extra.mcts) — the MCTS share of the step is measured on that shape",
                       "parallelism": "games sharded %d/GPU, no data-path collective" % G,
                       "whole_move_sims_per_s": (full_move or {}).get("sims_per_s"),
                       "hbm_used_bytes_per_rank": hbm_per_rank,
                       "tower_queues": 2 if two_queues else 1},
            "roofline": ({"bound": "hbm", "achieved": wino_detail["wino_gemm"]["achieved_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                          "frac": wino_detail["wino_gemm"]["achieved_GBs"] / HBM_PEAK_GBS, "traffic": traffic,
                          "traffic_source": (traffic_source + " (separate rocprofv3 --pmc passes over this kernel on this shape, scripts/r5_pmc_tower.sh: FETCH_SIZE x 2 (gfx950) + WRITE_SIZE; QUOTED from the committed pass, not collected in this run)") if traffic_source else None,
                          "measured_in_this_run": ["achieved", "frac", "avg_launch_ms", "launches", "block"],
                          "algorithmic_bytes_per_launch": wino_detail["wino_gemm"]["algorithmic_bytes"],
                          # the whole dual block (all its kernels, one-queue HIP events around the block): VERDICT r3 item 2
                          "block": {"algorithmic_bytes": wino_detail["block_algorithmic_bytes"], "avg_ms": wino_detail["block_avg_ms"],
                                    "achieved": wino_detail["block_achieved_GBs"], "unit": "GB/s",
                                    "frac": (wino_detail["block_achieved_GBs"] / HBM_PEAK_GBS) if wino_detail["block_achieved_GBs"] else None,
                                    "kernels": "wino_gemm_h2g_kernel + wino_oip_h2c_kernel (chained)" if wino_detail.get("chained") else "in + GEMM + out"},
                          # measured context for `frac` (scripts/probes/rw_probe.hip on 2 GB streams): what a bare streaming kernel with this
                          # kernel's byte mix (1 part read : 2 parts written, nothing re-used) reaches on this part, box to box — read from the
                          # committed probe results; not a claim about `peak`, which stays the guide's 8 TB/s
                          "stream_ceiling_same_mix_GBs": stream_ceiling, "stream_ceiling_source": stream_src,
                          "frac_of_stream_ceiling": ([wino_detail["wino_gemm"]["achieved_GBs"] / max(stream_ceiling), wino_detail["wino_gemm"]["achieved_GBs"] / min(stream_ceiling)]
                                                     if stream_ceiling else None)}
                         if (args.compute == "wino_h2" and wino_detail) else
                         {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": (achieved / peak) if achieved else None, "traffic": traffic, "traffic_source": traffic_source}) | {
                         # the SAME run in SURVEY 8(d)'s own units: direct-convolution-equivalent FLOPs of every evaluation / wall time, against the
                         # fp32 MFMA peak.  It EXCEEDS 1 by design — not skipped work (tests/test_headline_parity_gpu.py holds these outputs to the
                         # oracle at this shape) but an algorithmic reformulation: Winograd F(5x5,3x3) executes 4.1x fewer multiplies than the
                         # direct convolution the FLOP count prices, and they run on the fp16 pipe (3 MFMAs per fp32-grade product).  `frac` above
                         # is a different quantity: the dominant kernel's bytes against the HBM roof.
                         "end_to_end": {"direct_equivalent_tflops": evals_sum * flops_eval / t_max / 1e12,
                                        "frac_of_fp32_mfma_peak": evals_sum * flops_eval / t_max / 1e12 / (FP32_MFMA_PEAK_TFLOPS * world),
                                        "gflop_per_eval_direct_equivalent": flops_eval / 1e9, "fp32_mfma_peak_tflops": FP32_MFMA_PEAK_TFLOPS,
                                        "why_above_1": "Winograd F(5x5,3x3) executes 4.1x fewer multiplies than the direct 3x3 convolution this FLOP count "
                                                       "prices, on the fp16 matrix pipe (fp16x2 split, fp32 accumulate); same outputs within the oracle tolerance"},
                         # a WHOLE move of all games on rank 0 (begin_move + Budget simulations + end_move), measured before the timed region: the
                         # sturdier figure — `value`'s K steps carry one move boundary in K instead of one in Budget
                         "whole_move": ({"sims_per_s": full_move["sims_per_s"], "seconds": full_move["seconds"], "moves_per_s": full_move["moves_per_s"]}
                                        if full_move else None),
                         "kernel": kernels[args.compute], "peak_note": notes[args.compute],
                         "flops_per_launch": flops_launch, "avg_launch_ms": launch_ms,
                         "launches": n_launch,
                         "timing": ("HIP events on the launch stream around every %d-th launch of the kernel inside the timed region "
                                    "(%d launches bracketed)" % (args.prof_stride, n_launch)) if in_region_prof else
                                   ("HIP events on the launch stream around every launch of the kernel on 24 ONE-queue steps right after "
                                    "the timed region (%d launches): the timed region runs the tower on two queues (agz_net_set_tower_queues), "
                                    "where the half-batch chains overlap and a kernel's own duration is not defined; "
                                    "`bench.py --tower-queues 1` measures the same kernel inside the timed region" % n_launch) if two_queues else
                                   "HIP events on the launch stream, steps after the timed region (--prof-stride 0)"},
            # a WHOLE move of all games on this rank (begin_move + Budget simulations + end_move), measured before the timed region:
            # `value`'s K steps carry one move boundary (1 in K instead of 1 in Budget), so `value` reads a little low
            "full_move_sims_per_s": (full_move or {}).get("sims_per_s"),
            "full_move_note": "rank 0's whole move (extra.full_move_19x19): re-root / prepareRoot + %d simulations + bestMove / Apply for all %d games; not aggregated over ranks" % (args.budget, G),
            # N > 1 self-checks, top level: every rank's simulations, and what the RCCL exchange leg (extra.examples_allgather) saw
            "per_rank_sims": per_rank_sims, "world_size": world,
            "rccl_ranks": (gather or {}).get("rccl_ranks") if world > 1 else None,
            "rows_gathered": (gather or {}).get("rows_gathered") if world > 1 else None,
            "rows_per_rank": rows_per_rank,
            "rows_check": (None if world == 1 else "ok" if rows_per_rank and min(rows_per_rank) >= 0 and sum(rows_per_rank) == (gather or {}).get("rows_gathered")
                           else "MISMATCH: rows_gathered is not the sum of rows_per_rank"),
            "extra": {"nn_evals_per_s": evals_sum / t_max, "iterations_per_s": iters_sum / t_max,
                      "per_rank_sims": per_rank_sims, "world_size": world,
                      "timed_region": {"moves_finished": st1["moves_played"] - st0["moves_played"],
                                       "move_boundaries": move_n // 2,
                                       "mean_path_nodes": (st1["path_nodes"] - st0["path_nodes"]) / max(1, sims_all),
                                       "mean_children_per_select": (st1["children_read"] - st0["children_read"]) / max(1, (st1["path_nodes"] - st0["path_nodes"]) - sims_all),
                                       "opening_moves_mean": float(n_open.mean()), "opening_moves_max": int(n_open.max()),
                                       "tree_sims_before_timing": (0 if args.no_pregrow else max(0, args.budget - args.warmup - args.steps // 2)) + args.warmup,
                                       "note": "every game on its own random opening; trees pre-grown so the timed steps cross a move "
                                               "boundary (end_move + begin_move of all games inside the timed region)"},
                      "full_move_19x19": full_move,
                      "mcts": mcts_detail,
                      # (replaced below by moves/s / the MEASURED mean game length when the complete-games leg runs)
                      "games_per_s_19x19": ({"value": full_move["moves_per_s"] / (2 * hw), "moves_per_game_cap": 2 * hw,
                                             "note": "LOWER BOUND: measured moves/s of a whole move / the 2*M*N move cap (the complete-games leg did not run)"}
                                            if full_move else None),
                      "end_to_end_tflops": evals_sum * flops_eval / t_max / 1e12,
                      "end_to_end_frac_of_fp32_peak": evals_sum * flops_eval / t_max / 1e12 / (FP32_MFMA_PEAK_TFLOPS * world),
                      "kernel_classes": prof, "examples_allgather": gather_ms, "compute": args.compute,
                      "other_compute_modes": legs, "wino": wino_detail,
                      "tree_full": st1["tree_full"]},
        }
        if world == 1 and not args.no_complete_games_leg and (S, K) == (19, 256):
            try:
                cg = complete_19x19_leg(ctx, nets[0], G=G, budget=args.complete_games_budget, S=S)
                out["extra"]["complete_games_19x19"] = cg
                if full_move and cg["games_finished"]:
                    mean_len = cg["game_length"]["mean"]
                    out["extra"]["games_per_s_19x19"] = {
                        "value": full_move["moves_per_s"] / mean_len, "moves_per_s_measured": full_move["moves_per_s"], "mean_game_length_measured": mean_len,
                        "lower_bound_at_move_cap": full_move["moves_per_s"] / (2 * hw), "moves_per_game_cap": 2 * hw,
                        "note": "MEASURED moves/s of a whole 800-simulation move (full_move_19x19) / the MEASURED mean length of %d complete 19x19 games under "
                                "this network (extra.complete_games_19x19: every game played to Ended() once at %d simulations per move; termination mix "
                                "there).  Round 5 divided by the 2*M*N move cap (lower_bound_at_move_cap).  A complete game at 800 simulations per move "
                                "takes hours: its length is measured at the small Budget, its move rate at the full one" % (cg["games_finished"], args.complete_games_budget)}
            except Exception as e:
                out["extra"]["complete_games_19x19"] = {"error": repr(e)}
        if world == 1 and not args.no_deep_leg:
            try:
                # two points: a policy as peaked as a trained one (largest prior 0.5 on average, ordinary value head), and the stress point — priors
                # only (value output zeroed), largest prior 0.95: mean path ~ 1 / (1 - p) nodes, the deepest trees this family produces
                # (the round's first run also measured 0.8 / priors only: profiles/r06/bench_n1_first.json)
                dl = {"peaked_0.5": deep_tree_leg(ctx, S, K, L, G, args.budget, args.compute, 0.5, False),
                      "priors_only_0.95": deep_tree_leg(ctx, S, K, L, G, args.budget, args.compute, 0.95, True)}
                dl["headline_for_comparison"] = {"sims_per_s": (full_move or {}).get("sims_per_s"), "mean_path_nodes": mcts_detail["mean_path_nodes"],
                                                 "max_path_nodes": mcts_detail["max_path_nodes"],
                                                 "mean_children_per_select": mcts_detail["mean_children_per_select"],
                                                 "k_select_ms_per_step": mcts_detail["k_select"]["avg_ms"], "k_expand_ms_per_step": mcts_detail["k_expand"]["avg_ms"]}
                out["extra"]["deep_tree_leg"] = dl
            except Exception as e:
                out["extra"]["deep_tree_leg"] = {"error": repr(e)}
        if world == 1 and not args.no_games_leg:
            try:
                out["extra"]["games_leg"] = games_leg(ctx, args.compute)
            except Exception as e:
                out["extra"]["games_leg"] = {"error": repr(e)}
        if world == 1 and not args.no_go9_leg:
            try:
                out["extra"]["go9_leg"] = go9_leg(ctx)
            except Exception as e:
                out["extra"]["go9_leg"] = {"error": repr(e)}
        if world == 1 and not args.no_latency_leg:
            try:
                out["extra"]["latency_leg"] = latency_leg(ctx)
            except Exception as e:
                out["extra"]["latency_leg"] = {"error": repr(e)}
        if world == 1 and not args.no_train_leg:
            try:
                out["extra"]["train_leg"] = train_leg(ctx)
            except Exception as e:
                out["extra"]["train_leg"] = {"error": repr(e)}
        if world == 1 and not args.no_games_leg:
            try:
                out["extra"]["config0_leg"] = config0_leg()
            except Exception as e:
                out["extra"]["config0_leg"] = {"error": repr(e)}
        if not args.no_cpu_baseline:   # rank 0 at ANY world size: the other ranks wait in the closing barrier
            try:
                out["cpu_baseline"] = cpu_baseline(S, K, L, max_threads=args.cpu_threads or None, g19_seconds=args.cpu_baseline_seconds)
                out["cpu_baseline"]["ran_on"] = "rank 0 only, once, after the timed region (the other %d rank(s) idle in a barrier)" % (world - 1)
            except Exception as e:  # the oracle is only the baseline leg; never the measured path
                out["cpu_baseline"] = {"value": None, "unit": "sims/s", "cores": 1, "kind": "port", "sample": "failed: %r" % (e,)}
        # this process's own clock from argument parsing to the line (the legs outside the timed region included; imports excluded)
        out["extra"]["bench_wall_seconds"] = time.perf_counter() - t_process
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if gather_hung:
        os._exit(0)   # a thread is stuck inside a collective: no clean teardown possible (the line is out)
    if world > 1:
        # the closing rendezvous on HOST tensors (gloo): ranks 1 .. N-1 wait here while rank 0 times the CPU baseline — about a minute; a
        # device-side barrier would park a spinning RCCL kernel on every other GPU for that long
        adist.reduce_step_timing(0.0, [0.0], device=None)
        dist.destroy_process_group()
    arena.close()
    for n_ in nets:
        n_.close()
    ctx.close()


if __name__ == "__main__":
    main()
