import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi
import test_deep_tree_fuzz_gpu as T

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(7000 + seed)
c = T.draw(rng)
print(c)
cells = c["m"] * c["n"]
A_ = c["n"] if c["kind"] == capi.GAME_C4 else cells
plen = A_ + 1
one = T.peaked(plen, cells, c["peak"], c["value_amp"], c["among_empty"], c["enc"] == capi.ENC_TWOPLANE)
openings = [int(x) for x in rng.integers(0, max(1, cells // 3), size=c["G"])]
G = c["G"]
ctx = A.Ctx(0)
kw = dict(DumbPass=c["DumbPass"], ResignPercentage=c["ResignPercentage"], PUCT=c["PUCT"], RandomCount=c["RandomCount"], RandomTemperature=1.0, RandomMinVisits=0)
dev = A.Arena(ctx, c["kind"], c["m"], c["n"], c["k"], c["komi"], encoder=c["enc"], n_games=G, seed=11, Budget=c["budget"], PassPreference=c["PassPreference"], max_moves=3 * cells, **kw)
from test_deep_tree_gpu import batched
if c['lanes'] > 1:
    dev.set_parallel(c['lanes'])
f = batched(one, plen)
dev.set_inferencer_callback(0, f, plen); dev.set_inferencer_callback(1, f, plen)
ab = np.array([(g % 2) == 0 for g in range(G)], np.uint8)
dev.reset(ab)
dev.random_moves(np.asarray(openings, np.int32), 11)
orcs = []
for g in range(G):
    o = O.Arena(T.KINDS[c["kind"]], c["m"], c["n"], c["k"], c["komi"], enc=c["enc"], Budget=c["budget"], seed=11 + g, PassPreference=c["PassPreference"], max_moves=3 * cells, **kw)
    o.set_callback(0, one, plen); o.set_callback(1, one, plen)
    if c['lanes'] > 1:
        o.set_parallel(c['lanes'])
    o.begin(int(ab[g]))
    for _ in range(openings[g]):
        o.random_move(11, g)
    orcs.append(o)
prev = dev.stats()
for ply in range(c["plies"]):
    dev.begin_move(); dev.simulate(c["budget"]); dev.end_move(True)
    st = dev.stats()
    print("ply", ply, "dev sims", st["sims_total"] - prev["sims_total"], "nonnull", st["sims_nonnull"] - prev["sims_nonnull"], "evals", st["nn_evals"] - prev["nn_evals"], "maxpath", dev.max_path_nodes())
    prev = st
    for g, o in enumerate(orcs):
        _, st0 = o.state()
        agent = 0 if ((st0["to_move"] == O.BLACK) == bool(ab[g])) else 1
        before = o.tree_stats(agent)
        o.step(True)
        after = o.tree_stats(agent)
        omv, ovis, obs, opr = o.root_children(agent)
        dmv, dvis, dbs, dpr = dev.root_children(g, agent)
        ok = np.array_equal(dmv, omv) and np.array_equal(dvis, ovis) and np.array_equal(dbs.view(np.uint32), obs.view(np.uint32))
        print("  game", g, "agent", agent, "orc playouts", after["playouts"] - before["playouts"], "iters", after["iters"] - before["iters"], "evals", after["nn_evals"] - before["nn_evals"],
              "nodes", after["nodes"], "dev nodes", dev.tree_nodes(g, agent), "root vis sum", int(ovis.sum()), int(dvis.sum()), "OK" if ok else "MISMATCH", "hist", dev.history(g)[-1], o.history()[-1], "passes", st0["passes"])
        if not ok:
            print("   dev", dmv[:6], dvis[:6], dbs[:6]); print("   orc", omv[:6], ovis[:6], obs[:6])
