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
def mk(b, lanes):
    dev = A.Arena(ctx, capi.GAME_C4, 6, 7, 4, 0.0, encoder=capi.ENC_TWOPLANE, n_games=1, seed=11, Budget=b, PassPreference=2, max_moves=126)
    if lanes > 1: dev.set_parallel(lanes)
    dev.set_inferencer_callback(0, f, plen); dev.set_inferencer_callback(1, f, plen)
    dev.reset(np.array([1], np.uint8))
    dev.random_moves(np.array([2], np.int32), 11)
    return dev
# (a) lanes 2: 292 sims in rounds of two, then two rounds of ONE lane
dev = mk(294, 2); dev.begin_move(); dev.simulate(292); dev.simulate(1)
print("lanes2 292+1      ", dev.root_children(0, 0)[1].tolist()); dev.simulate(1); print("lanes2 292+1+1    ", dev.root_children(0, 0)[1].tolist()); dev.close()
# (b) lanes 2: 292 then one round of two
dev = mk(294, 2); dev.begin_move(); dev.simulate(292); print("lanes2 292        ", dev.root_children(0, 0)[1].tolist(), dev.root_children(0,0)[2].tolist()); dev.simulate(2); print("lanes2 292+2      ", dev.root_children(0, 0)[1].tolist()); dev.close()
# (c) sequential
dev = mk(294, 1); dev.begin_move(); dev.simulate(292); print("seq 292           ", dev.root_children(0, 0)[1].tolist()); dev.simulate(1); print("seq 293           ", dev.root_children(0, 0)[1].tolist()); dev.close()
# oracle lanes 2 at 293 (last round one lane)
for b in (292, 293, 294):
    o = O.Arena(O.C4, 6, 7, 4, 0.0, enc=O.ENC_TWOPLANE, Budget=b, seed=11, PassPreference=2, max_moves=126)
    o.set_callback(0, one, plen); o.set_callback(1, one, plen); o.set_parallel(2); o.begin(1)
    for _ in range(2): o.random_move(11, 0)
    o.step(True); print("oracle lanes2", b, o.root_children(0)[1].tolist())
