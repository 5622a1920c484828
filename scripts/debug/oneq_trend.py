"""GEMM duration (HIP events, one queue) over successive groups of steps of the headline workload: is the bench's after-region sample stable?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import agogo_amd as A
from agogo_amd import capi
import bench
ctx = A.Ctx(0)
S, K, L, G = 19, 256, 20, 512
net = A.Net(ctx, K, L, 2 * K, S, S, 18, S * S + 1, bn_mode=capi.BN_IDENTITY)
net.init_random(1337); bench.standard_bn_init(net); net.commit(); net.set_compute_mode(capi.COMPUTE_WINO_H2)
arena = A.Arena(ctx, capi.GAME_WQ, S, S, komi=7.5, encoder=capi.ENC_WQ, n_games=G, seed=1337, Budget=800)
arena.set_inferencer(0, capi.INF_NET, net); arena.set_inferencer(1, capi.INF_NET, net)
arena.reset()
arena.random_moves(np.random.default_rng(1337).integers(0, 217, size=G).astype(np.int32), 1337)
arena.begin_move(); arena.simulate(40); ctx.sync()
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
for q in (2, 1):
    net.set_tower_queues(q)
    arena.simulate(4); ctx.sync()
    out = []
    for grp in range(8):
        if mode == "all":
            ctx.prof_enable(True)
        else:
            ctx.prof_enable(True, classes=[capi.PROF_WINO_GEMM])
        arena.simulate(6); ctx.sync()
        ctx.prof_enable(False)
        n, ms = ctx.prof_read(capi.PROF_WINO_GEMM)
        out.append(round(ms / n, 4))
    print("queues", q, "events on", mode, "GEMM avg ms per group of 6 steps:", out)
