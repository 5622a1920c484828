"""GPU: the search on NARROW, DEEP trees — the regime a trained, peaked policy produces and the near-uniform
random-init priors of the headline never reach: long PUCT descents, in-tree two-pass terminals (mcts/search.go:227,269 -> combinedScore,
utils.go:62-67), the M*N depth cap with its null simulations and the stall that follows (search.go:211-215).

The inferencer is a HOST function on both sides — AGZ_INF_CALLBACK on the device (leaves of all games in one call), the oracle's per-leaf
callback — computing a peaked policy from the leaf's encoded planes alone (both receive the encoder's tensor bit for bit), so the
comparison is network-independent.  Bit-exact: children, visits, blackScores bits, prior bits, moves, examples.
"""
import zlib

import numpy as np
import pytest

import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi

pytestmark = pytest.mark.gpu


def f32bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def peaked_by_hash(plen, peak=0.8, value_amp=0.0):
    """policy: `peak` on one action chosen by a CRC of the planes, the rest spread evenly; value: 0 (priors alone steer the search) or a
    small CRC-derived number"""
    def one(planes_flat):
        h = zlib.crc32(np.ascontiguousarray(planes_flat, np.float32).tobytes())
        p = np.full(plen, (1.0 - peak) / (plen - 1), np.float32)
        p[h % plen] = peak
        v = np.float32(value_amp * (((h >> 8) % 2001) - 1000) / 1000.0)
        return p, float(v)
    return one


def pass_for_black_fill_for_white(cells):
    """TWOPLANE planes (plane 0: stones +1 / -1 / 0.001 empty, plane 1: the mover): Black always wants to pass, White the first empty
    point — White fills the board while Black passes: paths far longer than M*N plies (the depth cap), and once White has no legal
    point left both pass: in-tree two-pass terminals"""
    plen = cells + 1

    def one(planes_flat):
        x = np.asarray(planes_flat, np.float32).reshape(2, cells)
        black_to_move = x[1, 0] > 0
        p = np.full(plen, 0.05 / plen, np.float32)
        if black_to_move:
            p[cells] = 0.95
        else:
            empt = np.where(np.abs(x[0]) < 0.5)[0]
            p[int(empt[0]) if len(empt) else cells] = 0.95
        return p, 0.0
    return one


def batched(one, plen):
    def f(leaves):
        n = leaves["planes"].shape[0]
        pol = np.zeros((n, plen), np.float32)
        val = np.zeros(n, np.float32)
        for i in range(n):
            pol[i], val[i] = one(leaves["planes"][i].reshape(-1))
        return pol, val
    return f


def run(ctx, kind, okind, m, n, komi, enc, one, plen, G, budget, plies, openings, lanes=1, PassPreference=capi.DONT_PREFER_PASS, max_moves=0, k=0, max_nodes=0, pool_policy=None, **kw):
    dev = A.Arena(ctx, kind, m, n, k, komi, encoder=enc, n_games=G, seed=11, Budget=budget, PassPreference=PassPreference, max_moves=max_moves,
                  max_nodes=max_nodes, **kw)
    if lanes > 1:
        dev.set_parallel(lanes)
    f = batched(one, plen)
    dev.set_inferencer_callback(0, f, plen)
    dev.set_inferencer_callback(1, f, plen)
    if pool_policy is not None:
        dev.set_pool_policy(pool_policy)
    ab = np.array([(g % 2) == 0 for g in range(G)], np.uint8)
    dev.reset(ab)
    dev.random_moves(np.asarray(openings, np.int32), 11)
    orcs = []
    for g in range(G):
        o = O.Arena(okind, m, n, k, komi, enc=enc, Budget=budget, seed=11 + g, PassPreference=PassPreference, max_moves=max_moves, **kw)
        o.set_callback(0, one, plen)
        o.set_callback(1, one, plen)
        if lanes > 1:
            o.set_parallel(lanes)
        o.begin(int(ab[g]))
        for _ in range(int(openings[g])):
            o.random_move(11, g)
        np.testing.assert_array_equal(dev.history(g), o.history())
        orcs.append(o)
    alive = [True] * G
    for ply in range(plies):
        if not any(alive):
            break
        dev.begin_move()
        dev.simulate(budget)
        dev.end_move(True)
        for g, o in enumerate(orcs):
            if not alive[g]:
                continue
            _, st0 = o.state()
            agent = 0 if ((st0["to_move"] == O.BLACK) == bool(ab[g])) else 1
            cont = o.step(True)
            omv, ovis, obs, opr = o.root_children(agent)
            dmv, dvis, dbs, dpr = dev.root_children(g, agent)
            msg = "game %d ply %d" % (g, ply)
            np.testing.assert_array_equal(dmv, omv, err_msg=msg)
            np.testing.assert_array_equal(dvis, ovis, err_msg=msg)
            np.testing.assert_array_equal(f32bits(dbs), f32bits(obs), err_msg=msg)
            np.testing.assert_array_equal(f32bits(dpr), f32bits(opr), err_msg=msg)
            assert dev.history(g)[-1] == o.history()[-1], msg
            alive[g] = cont
    return dev, orcs


def test_depth_cap_and_in_tree_two_pass_terminals_on_a_small_board(ctx):
    """5x5 Go (M*N = 25: the depth cap).  Black passes, White fills: descents run past 25 plies — null simulations, then the stall of a
    deterministic search (q14) — and where White has no point left both sides pass inside the tree: terminal leaves scored by
    combinedScore, no evaluation.  Device == oracle after every ply; the run really reached both regimes."""
    m = 5
    one = pass_for_black_fill_for_white(m * m)
    dev, orcs = run(ctx, capi.GAME_WQ, O.WQ, m, m, 0.5, capi.ENC_TWOPLANE, one, m * m + 1, G=4, budget=400, plies=6, openings=[0, 3, 6, 9],
                    PassPreference=capi.DONT_PREFER_PASS)
    st = dev.stats()
    nulls = st["sims_total"] - st["sims_nonnull"]
    assert nulls > 0, st                                               # the depth cap was hit (search.go:211-215)
    print("\n[depth cap 5x5] sims %d, null %d, evals %d, mean path nodes %.1f" % (st["sims_total"], nulls, st["nn_evals"], st["path_nodes"] / st["sims_total"]))
    assert st["path_nodes"] / max(1, st["sims_nonnull"]) > 5, st       # long descents (measured: 6.5 nodes per simulation on a 25-cell board)
    tot = {"evals": 0, "playouts": 0}
    for o in orcs:
        for a in (0, 1):
            ts = o.tree_stats(a)
            tot["evals"] += ts["nn_evals"]
            tot["playouts"] += ts["playouts"]
    assert st["nn_evals"] == tot["evals"] and st["sims_nonnull"] == tot["playouts"]
    assert st["nn_evals"] < st["sims_nonnull"], st                     # simulations that ended in a two-pass terminal: backed up without an evaluation
    dev.close()


@pytest.mark.parametrize("lanes", [1, 4])
def test_narrow_deep_trees_on_9x9_with_the_wq_encoder(ctx, lanes):
    """9x9, WQ encoder (history planes), a policy with 0.95 on one CRC-chosen action and value 0: the search follows the priors — a fraction p of
    a node's visits continues along its best child wherever the chosen action is legal (measured: descents of up to 27 nodes) —
    re-rooting of deep narrow trees, also in lane rounds (stored virtual loss along long shared paths)."""
    S = 9
    one = peaked_by_hash(S * S + 1, 0.95)
    G = 6
    dev, orcs = run(ctx, capi.GAME_WQ, O.WQ, S, S, 5.5, capi.ENC_WQ, one, S * S + 1, G=G, budget=300 if lanes == 1 else 256, plies=5,
                    openings=[0, 5, 10, 15, 20, 30], lanes=lanes)
    st = dev.stats()
    print("\n[deep 9x9, lanes %d] mean path nodes %.1f, longest %d" % (lanes, st["path_nodes"] / st["sims_total"], dev.max_path_nodes()))
    # (measured: 4.7 nodes per descent on average — the CRC-chosen action is often an occupied point, the peak is then spread over the legal
    # moves — the longest 27; lane rounds of four: the longest 15)
    assert st["path_nodes"] / st["sims_total"] > 4, st["path_nodes"] / st["sims_total"]
    assert dev.max_path_nodes() >= (20 if lanes == 1 else 12)
    dp, dpol, dval, dgi = dev.examples()
    for g in range(G):
        ob, op, ov = orcs[g].examples()
        sel = dgi == g
        assert sel.sum() == ob.shape[0]
        if ob.shape[0]:
            np.testing.assert_array_equal(f32bits(dp[sel]), f32bits(ob))
            np.testing.assert_array_equal(f32bits(dpol[sel]), f32bits(op))
    dev.close()


def test_narrow_deep_trees_on_19x19_220_simulations(ctx):
    """19x19 (the headline board), 220 simulations per move, three plies from mid-game openings, policy peaked at 0.9 with a small value
    signal: deep paths on the big board, bit-exact against the oracle, pools never full."""
    S = 19
    one = peaked_by_hash(S * S + 1, 0.9, value_amp=0.05)
    dev, orcs = run(ctx, capi.GAME_WQ, O.WQ, S, S, 7.5, capi.ENC_WQ, one, S * S + 1, G=4, budget=220, plies=3, openings=[0, 60, 120, 200])
    st = dev.stats()
    assert st["tree_full"] == 0
    print("\n[deep 19x19] mean path nodes %.1f, longest %d, children read per level %.1f" % (st["path_nodes"] / st["sims_total"], dev.max_path_nodes(),
                                                                                         st["children_read"] / max(1, st["path_nodes"] - st["sims_total"])))
    assert st["path_nodes"] / st["sims_total"] > 4, st["path_nodes"] / st["sims_total"]
    assert dev.max_path_nodes() >= 12
    dev.close()


def test_puct_near_tie_needs_a_correctly_rounded_sqrt(ctx):
    """Round 6, found by the narrow-tree fuzz: Connect-4, a policy of 0.5 on one column and 1/14 on the others, value 0.  After 292 simulations
    the root's children stand at 153 / 21 / 21 ... visits: 0.5 * sqrt(300) / 154 against (1/14) * sqrt(300) / 22 — equal in exact arithmetic
    (0.5 / 154 = 1 / 308), two ulps apart in float32 in favour of the SECOND child when sqrt(300.f) is correctly rounded (0x418a9067), a tie — first
    child — when it is one ulp low (0x418a9066: what __fsqrt_rn AND sqrtf return on this toolchain for 15 % of the integers up to 2^24,
    scripts/probes/select_arith_probe.hip).  Simulation 293 is the witness; the sequential and the lane-round search both."""
    cells, plen = 42, 8

    def one(planes_flat):
        x = np.ascontiguousarray(planes_flat, np.float32)
        p = np.full(plen, 0.5 / 7, np.float32)
        empt = np.where(np.abs(x[:cells]) < 0.5)[0]
        p[zlib.crc32(x.tobytes()) % plen if plen <= cells else int(empt[0])] = 0.5
        return p, 0.0

    f = batched(one, plen)
    for lanes, budget in ((1, 292), (1, 293), (1, 294), (2, 293), (2, 294)):
        dev = A.Arena(ctx, capi.GAME_C4, 6, 7, 4, 0.0, encoder=capi.ENC_TWOPLANE, n_games=1, seed=11, Budget=budget, PassPreference=capi.DONT_RESIGN, max_moves=126)
        if lanes > 1:
            dev.set_parallel(lanes)
        dev.set_inferencer_callback(0, f, plen)
        dev.set_inferencer_callback(1, f, plen)
        dev.reset(np.array([1], np.uint8))
        dev.random_moves(np.array([2], np.int32), 11)
        o = O.Arena(O.C4, 6, 7, 4, 0.0, enc=O.ENC_TWOPLANE, Budget=budget, seed=11, PassPreference=capi.DONT_RESIGN, max_moves=126)
        o.set_callback(0, one, plen)
        o.set_callback(1, one, plen)
        if lanes > 1:
            o.set_parallel(lanes)
        o.begin(1)
        for _ in range(2):
            o.random_move(11, 0)
        dev.begin_move()
        dev.simulate(budget)
        dev.end_move(True)
        o.step(True)
        omv, ovis, obs, opr = o.root_children(0)
        dmv, dvis, dbs, dpr = dev.root_children(0, 0)
        np.testing.assert_array_equal(dmv, omv, err_msg="lanes %d budget %d" % (lanes, budget))
        np.testing.assert_array_equal(dvis, ovis, err_msg="lanes %d budget %d" % (lanes, budget))
        np.testing.assert_array_equal(f32bits(dpr), f32bits(opr))
        if (lanes, budget) == (1, 292):
            assert sorted(dvis.tolist()) == [21] * 7 + [153]             # the near-tie position itself
        if (lanes, budget) == (1, 293):
            assert sorted(dvis.tolist()) == [21] * 6 + [22, 153]         # the second child took simulation 293 (a low sqrt gives 154 / 21)
        dev.close()


def test_growing_pools_never_overflow_and_change_nothing(ctx):
    """AGZ_POOL_GROW.  13x13 Go, a 0.999-peaked policy, value 0: the tree keeps nearly all its nodes move after move and outgrows ANY fixed
    multiple of a search's expansions (the wide narrow-tree soak's one default-pool failure).  Two arenas on the same seeds: pools that
    start at ONE search's worth and grow, and strict pools sized for every search of the run — identical roots,
    histories and counters after every ply, all equal to the oracle's; the growing pools re-allocated at least twice, nothing overflowed."""
    S, G, budget, plies = 13, 4, 40, 10
    plen = S * S + 1
    one = peaked_by_hash(plen, 0.999)
    f = batched(one, plen)
    one_search = (budget + 2) * (plen + 1)
    arenas = []
    for policy, cap in ((capi.POOL_GROW, one_search), (capi.POOL_STRICT, (plies + 2) * one_search)):
        dev = A.Arena(ctx, capi.GAME_WQ, S, S, 0, 0.5, encoder=capi.ENC_TWOPLANE, n_games=G, seed=11, Budget=budget, max_nodes=cap, max_moves=3 * S * S)
        dev.set_inferencer_callback(0, f, plen)
        dev.set_inferencer_callback(1, f, plen)
        dev.set_pool_policy(policy)
        ab = np.array([(g % 2) == 0 for g in range(G)], np.uint8)
        dev.reset(ab)
        dev.random_moves(np.array([0, 4, 9, 15], np.int32), 11)
        arenas.append(dev)
    orcs = []
    for g in range(G):
        o = O.Arena(O.WQ, S, S, 0, 0.5, enc=O.ENC_TWOPLANE, Budget=budget, seed=11 + g, max_moves=3 * S * S)
        o.set_callback(0, one, plen)
        o.set_callback(1, one, plen)
        o.begin(int(ab[g]))
        for _ in range([0, 4, 9, 15][g]):
            o.random_move(11, g)
        orcs.append(o)
    for ply in range(plies):
        for dev in arenas:
            dev.begin_move()
            dev.simulate(budget // 2)          # (the move's simulations in two calls: the room was made for all of them at begin_move)
            dev.simulate(budget - budget // 2)
            dev.end_move(True)
        for g, o in enumerate(orcs):
            _, st0 = o.state()
            agent = 0 if ((st0["to_move"] == O.BLACK) == bool(ab[g])) else 1
            o.step(True)
            ref = o.root_children(agent)
            for dev in arenas:
                got = dev.root_children(g, agent)
                np.testing.assert_array_equal(got[0], ref[0], err_msg="game %d ply %d" % (g, ply))
                np.testing.assert_array_equal(got[1], ref[1], err_msg="game %d ply %d" % (g, ply))
                np.testing.assert_array_equal(f32bits(got[2]), f32bits(ref[2]))
                np.testing.assert_array_equal(f32bits(got[3]), f32bits(ref[3]))
                assert dev.history(g)[-1] == o.history()[-1]
    cap, grows = arenas[0].pool_capacity()
    nodes = max(arenas[0].tree_nodes(g, a) for g in range(G) for a in (0, 1))
    print("\n[growing pools] capacity %d -> %d nodes per pool in %d re-allocations; fullest tree %d nodes (one search's worth: %d)" % (one_search, cap, grows, nodes, one_search))
    assert grows >= 2 and cap > one_search and nodes > 2 * one_search
    assert arenas[1].pool_capacity() == ((plies + 2) * one_search, 0)
    for dev in arenas:
        assert dev.stats()["tree_full"] == 0
    # more simulations in a move than its Budget promised: room is made inside agz_arena_simulate
    dev = arenas[0]
    dev.begin_move()
    dev.simulate(5 * budget)
    dev.end_move(True)
    assert dev.stats()["tree_full"] == 0
    for dev in arenas:
        dev.close()
