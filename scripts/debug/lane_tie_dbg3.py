import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi
import test_deep_tree_fuzz_gpu as T
from test_deep_tree_gpu import batched
ctx = A.Ctx(0)
cells, plen = 42, 8
one = T.peaked(plen, cells, 0.5, 0.0, True, True)
f = batched(one, plen)
for lanes in (1, 2):
    for b in (292, 294):
        dev = A.Arena(ctx, capi.GAME_C4, 6, 7, 4, 0.0, encoder=capi.ENC_TWOPLANE, n_games=1, seed=11, Budget=b, PassPreference=2, max_moves=126)
        if lanes > 1: dev.set_parallel(lanes)
        dev.set_inferencer_callback(0, f, plen); dev.set_inferencer_callback(1, f, plen)
        dev.reset(np.array([1], np.uint8))
        dev.random_moves(np.array([2], np.int32), 11)
        o = O.Arena(O.C4, 6, 7, 4, 0.0, enc=O.ENC_TWOPLANE, Budget=b, seed=11, PassPreference=2, max_moves=126)
        o.set_callback(0, one, plen); o.set_callback(1, one, plen)
        if lanes > 1: o.set_parallel(lanes)
        o.begin(1)
        for _ in range(2): o.random_move(11, 0)
        dev.begin_move(); dev.simulate(b); dev.end_move(True)
        o.step(True)
        _, st = o.state()
        for agent in (0, 1):
            omv, ovis, obs, opr = o.root_children(agent)
            dmv, dvis, dbs, dpr = dev.root_children(0, agent)
            if len(omv):
                print("lanes", lanes, "b", b, "agent", agent, "vis dev", dvis.tolist(), "orc", ovis.tolist())
                print("   priors dev", [hex(x) for x in dpr.view(np.uint32)], "orc", [hex(x) for x in opr.view(np.uint32)])
        dev.close()
