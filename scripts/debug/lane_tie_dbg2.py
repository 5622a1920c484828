import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi
import test_deep_tree_fuzz_gpu as T
from test_deep_tree_gpu import batched

seed = 900924
rng = np.random.default_rng(7000 + seed)
c = T.draw(rng)
cells = 42; plen = 8
openings = [int(x) for x in rng.integers(0, max(1, cells // 3), size=c["G"])]
print(c, openings)
ctx = A.Ctx(0)
one = T.peaked(plen, cells, c["peak"], c["value_amp"], c["among_empty"], True)
f = batched(one, plen)
g = 2
for b in range(2, 302, 2):
    dev = A.Arena(ctx, capi.GAME_C4, 6, 7, 4, 0.0, encoder=capi.ENC_TWOPLANE, n_games=4, seed=11, Budget=b, PassPreference=2, max_moves=126)
    dev.set_parallel(2)
    dev.set_inferencer_callback(0, f, plen); dev.set_inferencer_callback(1, f, plen)
    ab = np.array([1, 0, 1, 0], np.uint8)
    dev.reset(ab)
    dev.random_moves(np.asarray(openings, np.int32), 11)
    o = O.Arena(O.C4, 6, 7, 4, 0.0, enc=O.ENC_TWOPLANE, Budget=b, seed=11 + g, PassPreference=2, max_moves=126)
    o.set_callback(0, one, plen); o.set_callback(1, one, plen)
    o.set_parallel(2)
    o.begin(int(ab[g]))
    for _ in range(openings[g]):
        o.random_move(11, g)
    _, st0 = o.state()
    agent = 0 if ((st0["to_move"] == O.BLACK) == bool(ab[g])) else 1
    dev.begin_move(); dev.simulate(b); dev.end_move(True)
    o.step(True)
    omv, ovis, obs, opr = o.root_children(agent)
    dmv, dvis, dbs, dpr = dev.root_children(g, agent)
    ok = np.array_equal(dmv, omv) and np.array_equal(dvis, ovis) and np.array_equal(dbs.view(np.uint32), obs.view(np.uint32))
    if not ok or b % 50 == 0 or b == 2:
        print(b, "OK" if ok else "MISMATCH", "to_move", st0["to_move"], "agent", agent, "dev", list(zip(dmv.tolist(), dvis.tolist())), "orc", list(zip(omv.tolist(), ovis.tolist())), np.round(dpr, 4).tolist())
    dev.close()
    if not ok:
        break
