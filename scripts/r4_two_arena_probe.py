"""Probe: the headline workload (19x19, K=256, 20 blocks, 512 concurrent games) as TWO arenas of 256 games on two contexts (streams),
their steps enqueued alternately, against ONE arena of 512 games with the tower on two queues (the bench's timed region).
What it asks: do the search kernels, input layer and heads of one half hide behind the other half's tower?"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.zeros(1, device="cuda")
import agogo_amd as A
from agogo_amd import capi

S, K, L = 19, 256, 20
STEPS = int(os.environ.get("PROBE_STEPS", "40"))


def make(ctx, G, seed, queues):
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
    net.set_tower_queues(queues)
    ar = A.Arena(ctx, capi.GAME_WQ, S, S, komi=7.5, encoder=capi.ENC_WQ, n_games=G, seed=seed, Budget=800, PUCT=1.0, RandomCount=0,
                 DumbPass=True, PassPreference=capi.DONT_PREFER_PASS)
    ar.set_inferencer(0, capi.INF_NET, net)
    ar.set_inferencer(1, capi.INF_NET, net)
    ar.reset()
    rng = np.random.default_rng(seed)
    ar.random_moves(rng.integers(0, int(0.6 * S * S) + 1, size=G).astype(np.int32), seed)
    ar.begin_move()
    ar.simulate(40)     # trees of some depth
    return net, ar


def run(label, arenas, ctxs):
    for _ in range(6):
        for a in arenas:
            a.simulate(1)
    for c in ctxs:
        c.sync()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        for a in arenas:
            a.simulate(1)
    for c in ctxs:
        c.sync()
    dt = time.perf_counter() - t0
    sims = STEPS * 512
    print(json.dumps({"form": label, "ms_per_512_sims": dt / STEPS * 1e3, "sims_per_s": sims / dt}), flush=True)


ctx0 = A.Ctx(0)
n1, a1 = make(ctx0, 512, 1337, 2)
run("one arena of 512 games, tower on two queues", [a1], [ctx0])
n1.set_tower_queues(1)
run("one arena of 512 games, one queue", [a1], [ctx0])
del a1, n1
ctxa, ctxb = A.Ctx(0), A.Ctx(0)
na, aa = make(ctxa, 256, 1337, 1)
nb, ab = make(ctxb, 256, 4242, 1)
run("two arenas of 256 games on two contexts, steps enqueued alternately", [aa, ab], [ctxa, ctxb])
run("the same, again", [aa, ab], [ctxa, ctxb])
