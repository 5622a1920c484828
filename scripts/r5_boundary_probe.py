"""Round-5 probe: what a move boundary of the headline workload costs (bench.py's 20-step region carries one): wall time of end_move,
begin_move (re-root + prepareRoot) and of the simulation steps right before / after, each fenced."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import agogo_amd as A
from agogo_amd import capi

ctx = A.Ctx(0)
S, K, L, G = 19, 256, int(os.environ.get("PROBE_L", "20")), int(os.environ.get("PROBE_G", "512"))
budget = int(os.environ.get("PROBE_BUDGET", "120"))
net = A.Net(ctx, K, L, 2 * K, S, S, 18, S * S + 1, bn_mode=capi.BN_IDENTITY)
net.init_random(1337)
for i in range(net.num_params()):
    name, n = net.param_info(i)
    if name.endswith("_gamma"):
        net.set_param(i, np.ones(n, np.float32))
    elif name.endswith("_beta"):
        net.set_param(i, np.zeros(n, np.float32))
net.commit()
net.set_compute_mode(capi.COMPUTE_WINO_H2)
arena = A.Arena(ctx, capi.GAME_WQ, S, S, komi=7.5, encoder=capi.ENC_WQ, n_games=G, seed=1337, Budget=budget, PUCT=1.0, RandomCount=0,
                DumbPass=True, PassPreference=capi.DONT_PREFER_PASS)
arena.set_inferencer(0, capi.INF_NET, net)
arena.set_inferencer(1, capi.INF_NET, net)
arena.reset()
rng = np.random.default_rng(1337)
arena.random_moves(rng.integers(0, int(0.6 * S * S) + 1, size=G).astype(np.int32), 1337)

def timed(f):
    ctx.sync(); t0 = time.perf_counter(); f(); ctx.sync(); return (time.perf_counter() - t0) * 1e3

out = []
for mv in range(4):
    s0 = arena.stats()
    tb = timed(arena.begin_move)
    s1 = arena.stats()
    first = [timed(lambda: arena.simulate(1)) for _ in range(3)]
    for _ in range(budget - 6):
        arena.simulate(1)
    last = [timed(lambda: arena.simulate(1)) for _ in range(3)]
    te = timed(lambda: arena.end_move(True))
    r = {"move": mv, "begin_move_ms": tb, "nn_evals_in_begin_move": s1["nn_evals"] - s0["nn_evals"], "first_steps_ms": first, "last_steps_ms": last, "end_move_ms": te}
    print(json.dumps(r), flush=True)
    out.append(r)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r5_boundary_probe.json", "w"), indent=1)
