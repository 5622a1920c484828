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
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
first = None
for b in range(2, 120, 2):
    dev = A.Arena(ctx, capi.GAME_C4, 6, 7, 4, 0.0, encoder=capi.ENC_TWOPLANE, n_games=1, seed=11, Budget=b, PassPreference=2, max_moves=126)
    dev.set_parallel(lanes)
    dev.set_inferencer_callback(0, f, plen); dev.set_inferencer_callback(1, f, plen)
    dev.reset(np.array([1], np.uint8))
    o = O.Arena(O.C4, 6, 7, 4, 0.0, enc=O.ENC_TWOPLANE, Budget=b, seed=11, PassPreference=2, max_moves=126)
    o.set_callback(0, one, plen); o.set_callback(1, one, plen)
    o.set_parallel(lanes)
    o.begin(1)
    dev.begin_move(); dev.simulate(b); dev.end_move(True)
    o.step(True)
    omv, ovis, obs, opr = o.root_children(0)
    dmv, dvis, dbs, dpr = dev.root_children(0, 0)
    ok = np.array_equal(dmv, omv) and np.array_equal(dvis, ovis) and np.array_equal(dbs.view(np.uint32), obs.view(np.uint32))
    print(b, "OK" if ok else "MISMATCH", "dev", list(zip(dmv.tolist(), dvis.tolist())), "orc", list(zip(omv.tolist(), ovis.tolist())), "priors", np.round(dpr, 4).tolist() if b == 2 else "")
    dev.close()
    if not ok and first is None:
        first = b
        break
