"""GPU: parity at BASELINE.json's sizes.

Where the oracle finishes in seconds it is compared directly; at the full 19x19 / K=256 / 20-block / 512-game
size the checks are size-independent properties: batch independence, determinism across identical games
(a checksum of checksums), MCTS conservation laws (search.go:392-408, tree.go:110, node.go:70-76), plus a
two-board spot check of the full-size network against the oracle.
"""
import os
import zlib

import numpy as np
import pytest

import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi

pytestmark = pytest.mark.gpu


def std_bn(net):
    for i in range(net.num_params()):
        name, n = net.param_info(i)
        if name.endswith("_gamma"):
            net.set_param(i, np.ones(n, np.float32))
        elif name.endswith("_beta"):
            net.set_param(i, np.zeros(n, np.float32))


def make_net(ctx, K, L, S, F, seed=1337):
    net = A.Net(ctx, K, L, 2 * K, S[1], S[0], F, (S[1] if F == 2 and S == (6, 7) else S[0] * S[1]) + 1,
                bn_mode=capi.BN_IDENTITY)
    net.init_random(seed)
    std_bn(net)
    net.commit()
    return net


def oracle_twin(net, K, L, S, F, Aspace):
    o = O.Net(K, L, 2 * K, S[1], S[0], F, Aspace, bn_mode=2)
    for i in range(net.num_params()):
        o.set_param(i, net.get_param(i))
    return o


def test_g19_network_full_size_spot_check_and_batch_independence(ctx):
    """config #4 network (K=256, 20 blocks, 19x19, B=512): 2 boards vs the oracle, and every board of the batch
    equals its own batch-1 evaluation (bit for bit with the latency regime off)."""
    S, K, L, F = (19, 19), 256, 20, 18
    net = make_net(ctx, K, L, S, F)
    rng = np.random.default_rng(1)
    x = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), size=(512, F, 19, 19), p=[0.2, 0.6, 0.2]).astype(np.float32)
    pol, val = net.infer(x)
    assert np.all(np.isfinite(pol)) and np.all(np.isfinite(val))
    np.testing.assert_allclose(pol.sum(axis=1), 1.0, atol=2e-5)
    for b in (0, 511, 257):
        # batch 1 takes the split-K latency regime: same maths, different fp32 summation order
        p1, v1 = net.infer(x[b:b + 1])
        np.testing.assert_allclose(p1[0], pol[b], atol=2e-5, rtol=2e-4)
        np.testing.assert_allclose(v1[0], val[b], atol=2e-4)
    net.set_latency_mode(False)  # regime off: one arithmetic at every batch size, bit for bit
    pol_off, val_off = net.infer(x)
    np.testing.assert_allclose(pol_off, pol, atol=2e-5, rtol=2e-4)   # (on: wide towers take the spread heads at every batch size)
    for b in (0, 511, 257):
        p1, v1 = net.infer(x[b:b + 1])
        np.testing.assert_array_equal(p1[0], pol_off[b])
        np.testing.assert_array_equal(v1[0], val_off[b])
    net.set_latency_mode(True)
    onet = oracle_twin(net, K, L, S, F, 362)
    po, vo = onet.infer(x[[0, 511]])
    np.testing.assert_allclose(pol[[0, 511]], po, atol=2e-5, rtol=2e-4)
    np.testing.assert_allclose(val[[0, 511]], vo, atol=2e-4)
    assert np.abs(pol[0] - pol[511]).max() > 1e-6


def _tree_digest(arena, g, agent):
    mv, vis, bs, pr = arena.root_children(g, agent)
    return zlib.crc32(mv.tobytes() + vis.tobytes() + bs.tobytes() + pr.tobytes())


def test_g19_512_games_conservation_and_determinism(ctx):
    """config #4 engine shape: 512 concurrent 19x19 games; synthetic hash inferencer; 3 plies x 24 sims."""
    G, budget = 512, 24
    ab = np.array([1, 0] * (G // 2), dtype=np.uint8)
    arena = A.Arena(ctx, capi.GAME_WQ, 19, 19, komi=7.5, encoder=capi.ENC_WQ, n_games=G, Budget=budget, max_nodes=40000)
    arena.set_inferencer(0, capi.INF_HASH)
    arena.set_inferencer(1, capi.INF_HASH)
    arena.reset(ab)
    orc = {}
    for ab_v in (1, 0):
        o = O.Arena(O.WQ, 19, 19, komi=7.5, enc=O.ENC_WQ, Budget=budget)
        o.set_inferencer(0, O.INF_HASH)
        o.set_inferencer(1, O.INF_HASH)
        o.begin(ab_v)
        orc[ab_v] = o
    for ply in range(3):
        s0 = arena.stats()
        arena.begin_move()
        arena.simulate(budget)
        arena.end_move(True)
        s1 = arena.stats()
        assert s1["sims_total"] - s0["sims_total"] == G * budget
        d_evals, d_sims = s1["nn_evals"] - s0["nn_evals"], s1["sims_nonnull"] - s0["sims_nonnull"]
        if ply < 2:   # fresh trees: prepareRoot evaluates every root once (search.go:392-408)
            assert d_evals == G + d_sims
        else:         # re-rooted trees (search.go:424-469): only roots that were still unexpanded are evaluated
            assert d_sims <= d_evals <= G + d_sims
        digests = {}
        for ab_v in (1, 0):
            o = orc[ab_v]
            _, st0 = o.state()
            agent = 0 if ((st0["to_move"] == O.BLACK) == bool(ab_v)) else 1
            o.step(True)
            omv, ovis, obs, opr = o.root_children(agent)
            g0 = 0 if ab_v == 1 else 1
            dmv, dvis, dbs, dpr = arena.root_children(g0, agent)
            np.testing.assert_array_equal(dmv, omv)
            np.testing.assert_array_equal(dvis, ovis)
            np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32))
            # conservation: every child starts at 1 visit; each non-null sim adds one visit to exactly one child
            assert int(dvis.sum()) - len(dvis) == budget
            assert abs(float(dpr.sum()) - 1.0) < 1e-4
            digests[ab_v] = (agent, _tree_digest(arena, g0, agent))
        # checksum of checksums: all 256 games with the same colour assignment are identical
        for g in range(0, G, 37):
            agent, want = digests[int(ab[g])]
            assert _tree_digest(arena, g, agent) == want, "game %d diverged at ply %d" % (g, ply)
            assert arena.history(g)[-1] == orc[int(ab[g])].history()[-1]
    assert arena.stats()["tree_full"] == 0


def _drive_with_gpu_net(ctx, kind, S, komi, enc, F, K, L, budget, plies, a_is_black, k=0):
    net = make_net(ctx, K, L, S, F)
    Aspace = net.conf.ActionSpace
    arena = A.Arena(ctx, kind, S[0], S[1], k, komi, encoder=enc, n_games=2, Budget=budget)
    arena.set_inferencer(0, capi.INF_NET, net)
    arena.set_inferencer(1, capi.INF_NET, net)
    ab = np.array(a_is_black, dtype=np.uint8)
    arena.reset(ab)

    def cb(planes):
        p, v = net.infer(planes.reshape(1, F, S[0], S[1]))
        return p[0], float(v[0])

    okind = {capi.GAME_C4: O.C4, capi.GAME_WQ: O.WQ}[kind]
    orcs = []
    for g in range(2):
        o = O.Arena(okind, S[0], S[1], k, komi, enc=enc, Budget=budget)
        o.set_callback(0, cb, Aspace)
        o.set_callback(1, cb, Aspace)
        o.begin(int(ab[g]))
        orcs.append(o)
    for ply in range(plies):
        arena.begin_move()
        arena.simulate(budget)
        arena.end_move(True)
        for g in range(2):
            o = orcs[g]
            _, st0 = o.state()
            if st0["ended"]:
                continue
            agent = 0 if ((st0["to_move"] == O.BLACK) == bool(ab[g])) else 1
            o.step(True)
            omv, ovis, obs, _ = o.root_children(agent)
            dmv, dvis, dbs, _ = arena.root_children(g, agent)
            np.testing.assert_array_equal(dmv, omv, err_msg="game %d ply %d" % (g, ply))
            np.testing.assert_array_equal(dvis, ovis, err_msg="game %d ply %d" % (g, ply))
            np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32))
            assert arena.history(g)[-1] == o.history()[-1]


def test_config2_connect4_k64_l6_400sims(ctx):
    """BASELINE config #2 (Connect-4, K=64, 6 blocks, 400 sims/move): first plies, device trees vs oracle trees fed
    with the same GPU network outputs."""
    _drive_with_gpu_net(ctx, capi.GAME_C4, (6, 7), 0.0, capi.ENC_TWOPLANE, 2, 64, 6, 400, 4, (1, 0), k=4)


def test_config3_go9_k128_l10_400sims(ctx):
    """BASELINE config #3 (9x9 Go, K=128, 10 blocks, 400 sims/move): first plies."""
    _drive_with_gpu_net(ctx, capi.GAME_WQ, (9, 9), 7.5, capi.ENC_WQ, 18, 128, 10, 400, 3, (1, 0))


def _drive_full_concurrency(ctx, kind, okind, S, komi, enc, F, K, L, G, budget, plies, mode, k=0, watch=(0, 1, 2)):
    """BASELINE configs[1]/[2] at their stated concurrency: G concurrent games (each on its own 0-6 move random opening so the
    batch is not G copies of one game), `plies` searched plies; the watched games' trees against oracle arenas whose inferencer
    is the same GPU network evaluated as a G-row batch of the one board (same kernels as the arena's batch)."""
    net = make_net(ctx, K, L, S, F)
    net.set_compute_mode(mode)
    Aspace = net.conf.ActionSpace
    arena = A.Arena(ctx, kind, S[0], S[1], k, komi, encoder=enc, n_games=G, Budget=budget)
    arena.set_inferencer(0, capi.INF_NET, net)
    arena.set_inferencer(1, capi.INF_NET, net)
    ab = np.array([(g % 2) == 0 for g in range(G)], dtype=np.uint8)
    arena.reset(ab)
    n_open = (np.arange(G) % 7).astype(np.int32)
    arena.random_moves(n_open, 4242)
    # batch independence of this arithmetic at this batch size (what lets the oracle's one-board callback stand for a batch row)
    rng = np.random.default_rng(3)
    x = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), size=(G, F, S[0], S[1])).astype(np.float32)
    p_all, _ = net.infer(x)
    p_rep, _ = net.infer(np.repeat(x[5:6], G, axis=0))
    np.testing.assert_array_equal(p_rep[G - 1], p_all[5])

    def cb(planes):
        p, v = net.infer(np.repeat(planes.reshape(1, F, S[0], S[1]), G, axis=0))
        return p[0], float(v[0])

    orcs = {}
    for g in watch:
        o = O.Arena(okind, S[0], S[1], k, komi, enc=enc, Budget=budget)
        o.set_callback(0, cb, Aspace)
        o.set_callback(1, cb, Aspace)
        o.begin(int(ab[g]))
        for _ in range(int(n_open[g])):
            o.random_move(4242, g)
        np.testing.assert_array_equal(arena.history(g), o.history())
        orcs[g] = o
    for ply in range(plies):
        arena.begin_move()
        arena.simulate(budget)
        arena.end_move(True)
        for g, o in orcs.items():
            _, st0 = o.state()
            if st0["ended"]:
                continue
            mover = O.BLACK if len(o.history()) % 2 == 0 else O.WHITE
            agent = 0 if ((mover == O.BLACK) == bool(ab[g])) else 1
            o.step(True)
            omv, ovis, obs, _ = o.root_children(agent)
            dmv, dvis, dbs, _ = arena.root_children(g, agent)
            np.testing.assert_array_equal(dmv, omv, err_msg="game %d ply %d" % (g, ply))
            np.testing.assert_array_equal(dvis, ovis, err_msg="game %d ply %d" % (g, ply))
            np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32))
            assert arena.history(g)[-1] == o.history()[-1]
    st = arena.stats()
    assert st["tree_full"] == 0 and st["examples_dropped"] == 0
    assert st["moves_played"] >= G * plies * 0.9
    # examples of the watched games (planes, one-hot policies) bit for bit
    dp, dq, dv, dg = arena.examples()
    for g, o in orcs.items():
        op, oq, ov = o.examples()
        rows = np.where(dg == g)[0]
        assert len(rows) == len(ov)
        for r, i in zip(rows, range(len(ov))):
            np.testing.assert_array_equal(dp[r].view(np.uint32), op[i].view(np.uint32))
            np.testing.assert_array_equal(dq[r], oq[i])
    arena.close()
    net.close()


@pytest.mark.parametrize("mode", [capi.COMPUTE_F32_MFMA, capi.COMPUTE_WINO_H2], ids=["f32", "wino_h2"])
def test_config2_connect4_256_concurrent_games_10_plies(ctx, mode):
    """BASELINE config #2 at its stated concurrency: Connect-4, K=64, 6 blocks, 256 concurrent games, 400 sims/move — in the
    default arithmetic and under the mode bench.py's games_leg sets (AGZ_COMPUTE_WINO_H2; at this shape, 84 row tiles on 256 CUs,
    the library keeps the fp32 half-tile kernels in that mode too: the leg's label names the MODE, this test pins what runs)."""
    _drive_full_concurrency(ctx, capi.GAME_C4, O.C4, (6, 7), 0.0, capi.ENC_TWOPLANE, 2, 64, 6, 256, 400, 10, mode, k=4,
                            watch=(0, 1, 130))


@pytest.mark.parametrize("mode", [capi.COMPUTE_BF16X3, capi.COMPUTE_WINO_H2], ids=["bf16x3", "wino_h2"])
def test_config3_go9_512_concurrent_games_10_plies(ctx, mode):
    """BASELINE config #3 at its stated concurrency: 9x9 Go, K=128, 10 blocks, 512 concurrent games, 400 sims/move, in the
    arithmetic bench.py's 9x9 leg (go9_leg) measures — AGZ_COMPUTE_WINO_H2: F(5x5,3x3), the chained block — and in bf16x3."""
    _drive_full_concurrency(ctx, capi.GAME_WQ, O.WQ, (9, 9), 7.5, capi.ENC_WQ, 18, 128, 10, 512, 400, 10, mode,
                            watch=(0, 3, 509))


@pytest.mark.parametrize("mode", [capi.COMPUTE_BF16X3, capi.COMPUTE_FP16X2, capi.COMPUTE_WINO, capi.COMPUTE_WINO_H2])
def test_engine_with_split_mode_network_bit_exact_vs_oracle(ctx, mode):
    """AGZ_COMPUTE_BF16X3 / AGZ_COMPUTE_FP16X2 under the engine: 32 concurrent 9x9 games (2592 GEMM rows: the throughput regime, bf16x3 dual
    blocks), device trees vs oracle trees.  The oracle's inferencer evaluates each leaf as a 32-row batch of the same
    board, so it takes the same kernel; the bf16x3 kernel is batch-independent bit for bit (checked first)."""
    S, K, L, F, G, budget = (9, 9), 128, 2, 18, 32, 24
    net = make_net(ctx, K, L, S, F)
    net.set_compute_mode(mode | capi.COMPUTE_FORCE)   # 2592 rows: below the chip-filling threshold
    rng = np.random.default_rng(4)
    x = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), size=(G, F, 9, 9)).astype(np.float32)
    p_all, v_all = net.infer(x)
    p_rep, v_rep = net.infer(np.repeat(x[5:6], G, axis=0))
    np.testing.assert_array_equal(p_rep[0], p_all[5])
    np.testing.assert_array_equal(p_rep[17], p_all[5])
    np.testing.assert_array_equal(v_rep[0], v_all[5])
    net.set_compute_mode(capi.COMPUTE_F32_MFMA)
    p_f32, _ = net.infer(x)
    assert not np.array_equal(p_f32, p_all)            # the engine below really runs the other arithmetic
    net.set_compute_mode(mode | capi.COMPUTE_FORCE)   # 2592 rows: below the chip-filling threshold

    arena = A.Arena(ctx, capi.GAME_WQ, 9, 9, 0, 7.5, encoder=capi.ENC_WQ, n_games=G, Budget=budget)
    arena.set_inferencer(0, capi.INF_NET, net)
    arena.set_inferencer(1, capi.INF_NET, net)
    ab = np.array([1, 0] * (G // 2), dtype=np.uint8)
    arena.reset(ab)

    def cb(planes):
        p, v = net.infer(np.repeat(planes.reshape(1, F, 9, 9), G, axis=0))
        return p[0], float(v[0])

    orcs = {}
    for g in (0, 1):
        o = O.Arena(O.WQ, 9, 9, 0, 7.5, enc=O.ENC_WQ, Budget=budget)
        o.set_callback(0, cb, 82)
        o.set_callback(1, cb, 82)
        o.begin(int(ab[g]))
        orcs[g] = o
    for ply in range(3):
        arena.begin_move()
        arena.simulate(budget)
        arena.end_move(True)
        for g, o in orcs.items():
            _, st0 = o.state()
            agent = 0 if ((st0["to_move"] == O.BLACK) == bool(ab[g])) else 1
            o.step(True)
            omv, ovis, obs, _ = o.root_children(agent)
            dmv, dvis, dbs, _ = arena.root_children(g, agent)
            np.testing.assert_array_equal(dmv, omv, err_msg="game %d ply %d" % (g, ply))
            np.testing.assert_array_equal(dvis, ovis, err_msg="game %d ply %d" % (g, ply))
            np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32))
            assert arena.history(g)[-1] == o.history()[-1]
        # games with the same colour assignment are identical (same net, same seedless search)
        for g in range(2, G, 5):
            assert np.array_equal(arena.history(g), arena.history(g % 2))


def test_complete_19x19_games_histories_labels_and_examples_match_the_oracle(ctx):
    """Arena.Play to Ended() on 19x19 (arena.go:96-155) — what bench.py's complete-games leg runs (small Budget, RandomCount
    16 with the per-tree RNG streams, DontPreferPass) — for 48 concurrent games; four watched games against oracle arenas played the same
    way: the whole move list (several hundred moves, randomised opening included), how the game ended, the winner, and every example row
    with its final label (planes, one-hot policy, value +1 / -1 / 0 from the winner: arena.go:146-155) bit for bit."""
    S, G, budget, seed = 19, 48, int(os.environ.get("AGZ_COMPLETE_BUDGET", "8")), 4242     # (soak knobs: Budget, watched games, oracle threads)
    kw = dict(Budget=budget, RandomCount=16, RandomMinVisits=1, RandomTemperature=1.0, DumbPass=True, PassPreference=capi.DONT_PREFER_PASS)
    dev = A.Arena(ctx, capi.GAME_WQ, S, S, 0, 7.5, encoder=capi.ENC_WQ, n_games=G, seed=seed, **kw)
    dev.set_inferencer(0, capi.INF_HASH)
    dev.set_inferencer(1, capi.INF_HASH)
    ab = np.array([(g % 2) == 0 for g in range(G)], np.uint8)
    dev.reset(ab)
    dev.play(0, record=True)
    st = dev.stats()
    assert st["games_finished"] == G and st["n_active"] == 0 and st["tree_full"] == 0
    lens = np.array([len(dev.history(g)) for g in range(G)])
    assert len({dev.history(g).tobytes() for g in range(G)}) == G            # every game its own (RandomCount)
    dp, dpol, dval, dgi = dev.examples()
    watch = (0, 13, 31, 47) if "AGZ_COMPLETE_WATCH" not in os.environ else tuple(range(0, G, max(1, G // int(os.environ["AGZ_COMPLETE_WATCH"]))))
    from concurrent.futures import ThreadPoolExecutor

    def oracle_game(g):
        o = O.Arena(O.WQ, S, S, 0, 7.5, enc=O.ENC_WQ, seed=seed + g, **kw)
        o.set_inferencer(0, O.INF_HASH)
        o.set_inferencer(1, O.INF_HASH)
        o.begin(int(ab[g]))
        o.play(0, True)
        return o

    with ThreadPoolExecutor(max_workers=min(int(os.environ.get("AGZ_HEADLINE_THREADS", "8")), os.cpu_count() or 2)) as ex:
        orcs = dict(zip(watch, ex.map(oracle_game, watch)))
    ends = {"two_passes": 0, "cap": 0, "other": 0}
    for g, o in orcs.items():
        oh = o.history()
        np.testing.assert_array_equal(dev.history(g), oh, err_msg="game %d" % g)
        _, ost = o.state()
        _, dst = dev.game(g)
        assert ost["ended"] and dst["ended"] and dst["winner"] == ost["winner"], g
        ob, op, ov = o.examples()
        sel = dgi == g
        assert sel.sum() == ob.shape[0] > 0
        np.testing.assert_array_equal(dp[sel].view(np.uint32), ob.view(np.uint32))
        np.testing.assert_array_equal(dpol[sel].view(np.uint32), op.view(np.uint32))
        np.testing.assert_array_equal(dval[sel], ov)
        assert set(np.unique(ov)) <= {-1.0, 0.0, 1.0}
        ends["two_passes" if (len(oh) >= 2 and oh[-1] == capi.PASS and oh[-2] == capi.PASS) else "cap" if len(oh) >= 2 * S * S else "other"] += 1
    print("\n[complete 19x19 games] Budget %d, %d watched of %d: lengths min %d mean %.1f max %d; watched endings %r" % (budget, len(watch), G, lens.min(), lens.mean(), lens.max(), ends))
    dev.close()
