"""Measured self-play games/s (complete games, not an estimate) on the BASELINE configs that finish in minutes:
  --config c4   configs[1]: Connect-4 7x6, K=64, 6 blocks, 256 concurrent games, 400 sims/move
  --config go9  configs[2]: 9x9 Go (wq), K=128, 10 blocks, 512 concurrent games, 400 sims/move
Continuous self-play (agz_arena_selfplay): a finished game's slot restarts at once, the run ends when `--games` games
have finished.  The 19x19 / 800-sim config needs ~3e8 simulations for 512 games (hours): bench.py reports its moves/s.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import agogo_amd as A
from agogo_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--config", choices=["c4", "go9"], default="go9")
ap.add_argument("--games", type=int, default=0, help="games to finish (default: the number of concurrent games)")
ap.add_argument("--compute", choices=["f32", "bf16x3", "fp16x2", "wino"], default="bf16x3")
args = ap.parse_args()
MODES = {"f32": capi.COMPUTE_F32_MFMA, "bf16x3": capi.COMPUTE_BF16X3, "fp16x2": capi.COMPUTE_FP16X2, "wino": capi.COMPUTE_WINO}
ctx = A.Ctx(0)
if args.config == "c4":
    K, L, G, sims = 64, 6, 256, 400
    net = A.Net(ctx, K, L, 2 * K, 7, 6, 2, 8, bn_mode=capi.BN_IDENTITY)
    mk = lambda: A.Arena(ctx, capi.GAME_C4, 6, 7, 4, encoder=capi.ENC_TWOPLANE, n_games=G, seed=1337, Budget=sims)
    name = "configs[1]: Connect-4 7x6, K=64, 6 blocks, 256 concurrent games, 400 sims/move"
else:
    K, L, G, sims = 128, 10, 512, 400
    net = A.Net(ctx, K, L, 2 * K, 9, 9, 18, 82, bn_mode=capi.BN_IDENTITY)
    mk = lambda: A.Arena(ctx, capi.GAME_WQ, 9, 9, 0, 7.5, encoder=capi.ENC_WQ, n_games=G, seed=1337, Budget=sims)
    name = "configs[2]: 9x9 Go (wq, komi 7.5, move cap 2*81), K=128, 10 blocks, 512 concurrent games, 400 sims/move"
net.init_random(1337)
for i in range(net.num_params()):   # standard BatchNorm initial state (bench.py)
    nm, n = net.param_info(i)
    if nm.endswith("_gamma"):
        net.set_param(i, [1.0] * n)
    elif nm.endswith("_beta"):
        net.set_param(i, [0.0] * n)
net.commit()
net.set_compute_mode(MODES[args.compute])
arena = mk()
arena.set_inferencer(0, capi.INF_NET, net)
arena.set_inferencer(1, capi.INF_NET, net)
arena.reset()
target = args.games or G
ctx.sync()
t0 = time.perf_counter()
arena.selfplay(target, record=True)
ctx.sync()
dt = time.perf_counter() - t0
st = arena.stats()
r = arena.results()
print(json.dumps({"workload": name, "compute": args.compute, "games_finished": st["games_finished"], "seconds": dt,
                  "games_per_s": st["games_finished"] / dt, "sims_per_s": st["sims_nonnull"] / dt,
                  "moves_per_s": st["moves_played"] / dt, "moves_per_game": st["moves_played"] / max(st["games_finished"], 1),
                  "examples": st["examples"], "results": r, "tree_full": st["tree_full"]}))
