"""GPU: parity AT the configurations and IN the arithmetic bench.py measures.

(i)   the headline network (19x19, K=256, 20 dual blocks, 512 boards — BASELINE configs[3] per GPU) in every compute mode the
      bench can select, on mid-game positions (random legal prefixes of 0..216 moves), against the oracle on a sample of
      boards: same tolerance as every other network test, max |delta| printed;
(ii)  the engine under that arithmetic at the headline shape: 512 concurrent 19x19 games on their own positions, device trees
      bit-exact against oracle trees fed the same network outputs;
(iii) BASELINE configs[4] (tournament Agent.Search: 40 blocks, ONE tree): the batch-1 network against the oracle, and a
      one-tree search — sequential and in lane rounds of 8 / 16 — bit-exact against the oracle's search.
"""
import os

import numpy as np
import pytest

import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi
from test_net_gpu import POL_ATOL, POL_RTOL, VAL_ATOL

pytestmark = pytest.mark.gpu

S, K, F, ASPACE = 19, 256, 18, 362
MODES = {"wino": capi.COMPUTE_WINO, "bf16x3": capi.COMPUTE_BF16X3, "fp16x2": capi.COMPUTE_FP16X2, "f32": capi.COMPUTE_F32_MFMA}
if hasattr(capi, "COMPUTE_WINO_H2"):
    MODES["wino_h2"] = capi.COMPUTE_WINO_H2


def std_net(ctx, L, seed=1337):
    net = A.Net(ctx, K, L, 2 * K, S, S, F, ASPACE, bn_mode=capi.BN_IDENTITY)
    net.init_random(seed)
    for i in range(net.num_params()):
        name, n = net.param_info(i)
        if name.endswith("_gamma"):
            net.set_param(i, np.ones(n, np.float32))
        elif name.endswith("_beta"):
            net.set_param(i, np.zeros(n, np.float32))
    net.commit()
    return net


def oracle_twin(net, L):
    o = O.Net(K, L, 2 * K, S, S, F, ASPACE, bn_mode=2)
    for i in range(net.num_params()):
        o.set_param(i, net.get_param(i))
    return o


def midgame_planes(B, seed=1337, most=216):
    """WQEncoder planes of B positions: slot b plays u ~ U[0, most] uniformly random legal moves from the empty board
    (SURVEY 8(d)), through the oracle's game rules"""
    out = np.zeros((B, F, S, S), np.float32)
    depth = np.zeros(B, np.int32)
    for b in range(B):
        rng = np.random.default_rng(seed + b)
        u = int(rng.integers(0, most + 1))
        g = O.Game(O.WQ, S, S, 0, 7.5)
        player = O.BLACK
        g.set_to_move(player)
        for _ in range(u):
            empt = np.where(g.board() == 0)[0]
            mv = -1
            for cand in rng.permutation(empt)[:8]:
                if g.check(player, int(cand)):
                    mv = int(cand)
                    break
            if mv < 0:
                break
            g.apply(player, mv)
            player = O.WHITE if player == O.BLACK else O.BLACK
            g.set_to_move(player)
            depth[b] += 1
        out[b] = g.encode(O.ENC_WQ).reshape(F, S, S)
    return out, depth


_CACHE = {}


def cached_midgame(B):
    if B not in _CACHE:
        _CACHE[B] = midgame_planes(B)
    return _CACHE[B]


@pytest.mark.parametrize("mode", list(MODES))
def test_headline_network_l20_b512_midgame_boards_vs_oracle(ctx, mode):
    """(i) K=256, L=20, B=512, 19x19 in the measured arithmetic: 32 boards spread evenly over the opening depths (the earliest and
    the deepest position included) against the oracle — one oracle thread per board, the oracle network is read-only — every
    board of the batch finite and normalised, and the batch agrees with the default fp32-MFMA arithmetic within the same tolerance."""
    L, B = 20, 512
    net = std_net(ctx, L)
    x, depth = cached_midgame(B)
    assert depth.max() >= 150 and len(np.unique(depth)) > 100
    p32, v32 = net.infer(x)
    net.set_compute_mode(MODES[mode])
    pol, val = net.infer(x)
    assert np.all(np.isfinite(pol)) and np.all(np.isfinite(val))
    np.testing.assert_allclose(pol.sum(axis=1), 1.0, atol=2e-5)
    if mode != "f32":
        assert not np.array_equal(pol, p32), "the mode under test did not change the arithmetic"
    np.testing.assert_allclose(pol, p32, atol=POL_ATOL, rtol=POL_RTOL)
    np.testing.assert_allclose(val, v32, atol=VAL_ATOL)
    order = np.argsort(depth, kind="stable")
    pick = [int(order[i]) for i in np.linspace(0, B - 1, 32).round().astype(int)]
    key = ("oracle", L, tuple(pick))
    if key not in _CACHE:   # the same net (seed 1337) and boards in every mode: the oracle runs once, one thread per board
        from concurrent.futures import ThreadPoolExecutor
        onet = oracle_twin(net, L)
        with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 4)) as ex:
            res = list(ex.map(lambda b: onet.infer(x[b:b + 1]), pick))
        _CACHE[key] = (np.concatenate([r[0] for r in res]), np.concatenate([r[1] for r in res]))
    po, vo = _CACHE[key]
    dp, dv = np.abs(pol[pick] - po).max(), np.abs(val[pick] - vo).max()
    print("\n[headline parity] mode=%s %d boards vs the oracle (moves played %d..%d): max|dpolicy|=%.3g max|dvalue|=%.3g (tolerance %g + %g*|p|, %g)"
          % (mode, len(pick), depth[pick].min(), depth[pick].max(), dp, dv, POL_ATOL, POL_RTOL, VAL_ATOL))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "headline_parity.txt"), "a") as f:
            f.write("mode=%s oracle_boards=%d depth=%d..%d max_dpolicy=%.4g max_dvalue=%.4g max_policy=%.4g\n"
                    % (mode, len(pick), depth[pick].min(), depth[pick].max(), dp, dv, float(po.max())))
    np.testing.assert_allclose(pol[pick], po, atol=POL_ATOL, rtol=POL_RTOL)
    np.testing.assert_allclose(val[pick], vo, atol=VAL_ATOL)
    assert np.abs(pol[pick[0]] - pol[pick[-1]]).max() > 1e-6
    # batch independence of the measured arithmetic: a board evaluated as 512 copies of itself gives the same bits
    rep, vrep = net.infer(np.repeat(x[pick[20]:pick[20] + 1], B, axis=0))
    np.testing.assert_array_equal(rep[0], pol[pick[20]])
    np.testing.assert_array_equal(rep[B - 1], pol[pick[20]])
    np.testing.assert_array_equal(vrep[7], val[pick[20]])
    net.close()


@pytest.mark.parametrize("mode", ["wino"] + (["wino_h2"] if "wino_h2" in MODES else []))
def test_headline_engine_512_games_on_their_own_positions_bit_exact(ctx, mode):
    """(ii) configs[3] per GPU as bench.py runs it: 512 games, each on its own random opening, the measured arithmetic,
    two searched plies; 32 watched games against oracle arenas whose inferencer is the same GPU network evaluated as a batch of copies
    of the one board (agz_net_min_same_batch rows: the kernels of the 512-row batch; the arithmetic is batch independent bit for bit)."""
    L, G, budget, seed = 20, 512, 12, 1337
    net = std_net(ctx, L)
    net.set_compute_mode(MODES[mode])
    dev = A.Arena(ctx, capi.GAME_WQ, S, S, 0, 7.5, encoder=capi.ENC_WQ, n_games=G, seed=seed, Budget=budget, max_nodes=12000)
    dev.set_inferencer(0, capi.INF_NET, net)
    dev.set_inferencer(1, capi.INF_NET, net)
    ab = np.array([(g % 2) == 0 for g in range(G)], dtype=np.uint8)
    dev.reset(ab)
    rng = np.random.default_rng(seed)
    n_moves = rng.integers(0, 217, size=G).astype(np.int32)
    n_moves[0], n_moves[1] = 216, 40
    dev.random_moves(n_moves, seed)
    assert len({dev.game(g)[0].tobytes() for g in range(0, G, 16)}) == G // 16

    nb = net.min_same_batch(1, G)      # the smallest batch with the G-board batch's kernels: per board the same bits (asserted below)

    def cb(planes):
        p, v = net.infer(np.repeat(planes.reshape(1, F, S, S), nb, axis=0))
        return p[0], float(v[0])

    x0 = cached_midgame(512)[0][7:8]
    pa, va = net.infer(np.repeat(x0, nb, axis=0))
    pb, vb = net.infer(np.repeat(x0, G, axis=0))
    np.testing.assert_array_equal(pa[0], pb[G - 1])
    np.testing.assert_array_equal(va[0], vb[0])
    # 32 watched games (round 5: 3) spread over the arena — both colour assignments, openings from 0 to 216 moves
    watch = tuple(sorted(set([0, 1, 311] + list(range(7, G, 17))[:29])))
    assert len(watch) == 32
    orcs = {}
    for g in watch:
        o = O.Arena(O.WQ, S, S, 0, 7.5, enc=O.ENC_WQ, Budget=budget, seed=seed + g)
        o.set_callback(0, cb, ASPACE)
        o.set_callback(1, cb, ASPACE)
        o.begin(int(ab[g]))
        for _ in range(int(n_moves[g])):
            o.random_move(seed, g)
        np.testing.assert_array_equal(dev.history(g), o.history())
        orcs[g] = o
    for ply in range(2):
        dev.begin_move()
        dev.simulate(budget)
        dev.end_move(True)
        for g, o in orcs.items():
            _, st0 = o.state()
            agent = 0 if ((st0["to_move"] == O.BLACK) == bool(ab[g])) else 1
            o.step(True)
            omv, ovis, obs, _ = o.root_children(agent)
            dmv, dvis, dbs, _ = dev.root_children(g, agent)
            np.testing.assert_array_equal(dmv, omv, err_msg="game %d ply %d" % (g, ply))
            np.testing.assert_array_equal(dvis, ovis, err_msg="game %d ply %d" % (g, ply))
            np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32))
            assert dev.history(g)[-1] == o.history()[-1]
    st = dev.stats()
    assert st["tree_full"] == 0 and st["sims_total"] == 2 * G * budget
    dev.close()
    net.close()


def test_headline_engine_deep_tree_across_a_re_root_bit_exact(ctx):
    """(ii-b) VERDICT r3 item 4: ONE watched game of the 512 under the measured arithmetic (AGZ_COMPUTE_WINO_H2, the chained block)
    with a deep tree — Budget 220 simulations per move from a 90-move position, three searched plies, so the third ply searches a
    tree RE-ROOTED from the agent's first search (mcts/search.go:424-500; prepareRoot / pipeline :209-257) — against an oracle arena
    fed the same network outputs (the GPU network evaluated as a 512-row batch of the one board: same kernels, batch-independent
    bit for bit).  Children, visit counts, blackScores and prior bits of the searching agent's root after every ply, and the moves played."""
    L, G, budget, seed, g = 20, 512, 220, 77, 5
    net = std_net(ctx, L)
    net.set_compute_mode(MODES["wino_h2"])
    dev = A.Arena(ctx, capi.GAME_WQ, S, S, 0, 7.5, encoder=capi.ENC_WQ, n_games=G, seed=seed, Budget=budget, max_nodes=120000)
    dev.set_inferencer(0, capi.INF_NET, net)
    dev.set_inferencer(1, capi.INF_NET, net)
    ab = np.array([(i % 2) == 1 for i in range(G)], dtype=np.uint8)
    dev.reset(ab)
    rng = np.random.default_rng(seed)
    n_moves = rng.integers(0, 217, size=G).astype(np.int32)
    n_moves[g] = 90
    dev.random_moves(n_moves, seed)

    def cb(planes):
        p, v = net.infer(np.repeat(planes.reshape(1, F, S, S), G, axis=0))
        return p[0], float(v[0])

    o = O.Arena(O.WQ, S, S, 0, 7.5, enc=O.ENC_WQ, Budget=budget, seed=seed + g)
    o.set_callback(0, cb, ASPACE)
    o.set_callback(1, cb, ASPACE)
    o.begin(int(ab[g]))
    for _ in range(int(n_moves[g])):
        o.random_move(seed, g)
    np.testing.assert_array_equal(dev.history(g), o.history())
    nodes = []
    for ply in range(3):
        dev.begin_move()
        dev.simulate(budget)
        dev.end_move(True)
        _, st0 = o.state()
        agent = 0 if ((st0["to_move"] == O.BLACK) == bool(ab[g])) else 1
        o.step(True)
        omv, ovis, obs, opr = o.root_children(agent)
        dmv, dvis, dbs, dpr = dev.root_children(g, agent)
        np.testing.assert_array_equal(dmv, omv, err_msg="ply %d" % ply)
        np.testing.assert_array_equal(dvis, ovis, err_msg="ply %d" % ply)
        np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32))
        np.testing.assert_array_equal(dpr.view(np.uint32), opr.view(np.uint32))
        if ply == 0:
            first_root_visits = int(ovis.sum())
        if ply == 2:   # the re-rooted search: its root carries the visits of the subtree kept from ply 0 on top of this search's
            assert int(ovis.sum()) > first_root_visits, (int(ovis.sum()), first_root_visits)
        assert dev.history(g)[-1] == o.history()[-1]
        nodes.append(dev.tree_nodes(g, agent))
    st = dev.stats()
    assert st["tree_full"] == 0
    assert min(nodes) > 10000, nodes          # 220 expansions x up to 362 children: deep pools, well past one block of children
    dev.close()
    net.close()


@pytest.mark.parametrize("mode", ["wino_h2", "f32"])
def test_prepare_root_packed_batch_equals_the_whole_batch(ctx, mode):
    """Round 5: prepareRoot (mcts/search.go:392-408) evaluates only the roots without children — with tree reuse a minority of the 512
    games per move boundary — as a PACKED batch of the smallest size that takes the kernels of the whole-arena batch
    (agz_net::min_same_batch), instead of all 512 boards (12 ms of a move boundary at L = 20).  Two arenas on the same seeds, one with the
    packed forward, one with the whole-batch forward (agz_arena_set_prep_compact, agz_debug.h): every watched root's children, visit
    counts, blackScores bits and prior bits after each of four searched plies, every game's history, and the statistics must be
    identical; the packed batch must really be smaller, and cover the roots that needed the network."""
    L, G, budget, seed = 3, 512, 12, 31
    net = std_net(ctx, L)
    net.set_compute_mode(MODES[mode])
    arenas = []
    for packed in (True, False):
        dev = A.Arena(ctx, capi.GAME_WQ, S, S, 0, 7.5, encoder=capi.ENC_WQ, n_games=G, seed=seed, Budget=budget, max_nodes=20000)
        dev.set_inferencer(0, capi.INF_NET, net)
        dev.set_inferencer(1, capi.INF_NET, net)
        dev.reset(np.array([(i % 2) == 1 for i in range(G)], dtype=np.uint8))
        rng = np.random.default_rng(seed)
        dev.random_moves(rng.integers(0, 200, size=G).astype(np.int32), seed)
        dev.set_prep_compact(packed)
        arenas.append(dev)
    packed_batches = []
    for ply in range(4):
        for dev in arenas:
            dev.begin_move()
        b1, r1 = arenas[0].last_prep_batch()
        b0, r0 = arenas[1].last_prep_batch()
        assert r1 == r0                                   # the same roots need the network
        assert b0 in (0, G) and (b0 == 0) == (r0 == 0)    # the whole-batch arena: all boards, or no forward at all
        assert (b1 == 0) == (r1 == 0) and r1 <= b1 <= G
        packed_batches.append((b1, r1))
        for dev in arenas:
            dev.simulate(budget)
            dev.end_move(True)
        for g in list(range(0, G, 37)) + [G - 1]:
            for agent in (0, 1):
                a, b = arenas[0].root_children(g, agent), arenas[1].root_children(g, agent)
                np.testing.assert_array_equal(a[0], b[0], err_msg="game %d agent %d ply %d" % (g, agent, ply))
                np.testing.assert_array_equal(a[1], b[1])
                np.testing.assert_array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
                np.testing.assert_array_equal(a[3].view(np.uint32), b[3].view(np.uint32))
            np.testing.assert_array_equal(arenas[0].history(g), arenas[1].history(g))
    s1, s0 = arenas[0].stats(), arenas[1].stats()
    for k in ("sims_total", "sims_nonnull", "nn_evals", "moves_played", "examples", "tree_full"):
        assert s1[k] == s0[k], k
    # ply 0: no tree yet, every root needs the network (whole batch); from ply 2 on (both agents have searched once) trees are re-used
    assert packed_batches[0] == (G, G), packed_batches
    assert any(0 < r < G and b < G for b, r in packed_batches[2:]), packed_batches
    print("prepareRoot batches (boards, roots):", packed_batches)
    for dev in arenas:
        dev.close()
    net.close()


@pytest.mark.parametrize("kind,m,n,k,enc,K,L,G,budget,plies", [
    (capi.GAME_MNK, 3, 3, 3, capi.ENC_TWOPLANE, 32, 2, 37, 200, 6),        # tic-tac-toe: games END inside the run (restarts, fresh roots)
    (capi.GAME_C4, 6, 7, 4, capi.ENC_TWOPLANE, 64, 3, 100, 150, 6),        # connect-4 at a batch where the half-tile fp32 kernels run
    (capi.GAME_WQ, 9, 9, 0, capi.ENC_WQ, 128, 3, 300, 160, 5),             # 9x9 Go, the chained Winograd tower at 300 boards
    (capi.GAME_MNK, 5, 5, 4, capi.ENC_TWOPLANE, 32, 1, 9, 90, 6),          # a handful of games: the packed batch may be 1..9 boards
], ids=["ttt", "c4", "go9", "mnk5"])
def test_prepare_root_packed_batch_on_other_games_and_sizes(ctx, kind, m, n, k, enc, K, L, G, budget, plies):
    """The packed prepareRoot batch against the whole-batch forward across games, network widths and arena sizes — wherever
    agz_net::min_same_batch finds a smaller batch with the whole batch's kernels the trees must stay identical, wherever it does not the
    forward must simply be the whole batch.  AGZ_COMPUTE_AUTO (what a user gets), every game's root statistics after every ply."""
    F = 18 if enc == capi.ENC_WQ else 2
    A1 = n + 1 if kind == capi.GAME_C4 else m * n + 1      # (c4: one entry per column + pass, as bench.py's config #2 net)
    net = A.Net(ctx, K, L, 64, n, m, F, A1, bn_mode=capi.BN_IDENTITY)
    net.init_random(7)
    for i in range(net.num_params()):
        name, cnt = net.param_info(i)
        if name.endswith("_gamma"):
            net.set_param(i, np.ones(cnt, np.float32))
        elif name.endswith("_beta"):
            net.set_param(i, np.zeros(cnt, np.float32))
    net.commit()
    net.set_compute_mode(capi.COMPUTE_AUTO)
    arenas = []
    for packed in (True, False):
        dev = A.Arena(ctx, kind, m, n, k, 5.5 if kind in (capi.GAME_WQ, capi.GAME_KOMI) else 0.0, encoder=enc, n_games=G, seed=99, Budget=budget)
        dev.set_inferencer(0, capi.INF_NET, net)
        dev.set_inferencer(1, capi.INF_NET, net)
        dev.reset()
        # every game on its own position (identical games would need the network all together or not at all)
        dev.random_moves(np.random.default_rng(5).integers(0, max(2, (n if kind == capi.GAME_C4 else m * n) // 3), size=G).astype(np.int32), 5)
        dev.set_prep_compact(packed)
        arenas.append(dev)
    seen_packed, batches = False, []
    for ply in range(plies):
        for dev in arenas:
            dev.begin_move()
        (b1, r1), (b0, r0) = arenas[0].last_prep_batch(), arenas[1].last_prep_batch()
        assert r1 == r0 and b0 in (0, G) and (b1 == 0) == (r1 == 0) and r1 <= b1 <= G, (ply, b1, r1, b0, r0)
        seen_packed = seen_packed or (0 < b1 < G)
        batches.append((b1, r1))
        for dev in arenas:
            dev.simulate(budget)
            dev.end_move(True)
        for g in range(G):
            for agent in (0, 1):
                a, b = arenas[0].root_children(g, agent), arenas[1].root_children(g, agent)
                np.testing.assert_array_equal(a[0], b[0], err_msg="game %d agent %d ply %d" % (g, agent, ply))
                np.testing.assert_array_equal(a[1], b[1])
                np.testing.assert_array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
                np.testing.assert_array_equal(a[3].view(np.uint32), b[3].view(np.uint32))
            np.testing.assert_array_equal(arenas[0].history(g), arenas[1].history(g))
    s1, s0 = arenas[0].stats(), arenas[1].stats()
    for key in ("sims_total", "sims_nonnull", "nn_evals", "moves_played", "games_finished", "examples", "tree_full"):
        assert s1[key] == s0[key], key
    print("packed batch used:", seen_packed, "batches", batches)
    for dev in arenas:
        dev.close()
    net.close()


def test_config4_l40_network_batch1_and_lane_batches_vs_oracle(ctx):
    """(iii-a) BASELINE configs[4] tower (40 blocks): batch 1 (the split-K latency regime), batches 8 and 16 (one lane round)
    against the oracle on three mid-game boards."""
    L = 40
    net = std_net(ctx, L)
    x, depth = cached_midgame(512)
    pick = [int(np.argmax(depth)), int(np.argsort(depth)[256]), int(np.argmin(depth))]
    onet = oracle_twin(net, L)
    po, vo = onet.infer(x[pick])
    worst = 0.0
    for i, b in enumerate(pick):
        p1, v1 = net.infer(x[b:b + 1])
        np.testing.assert_allclose(p1[0], po[i], atol=POL_ATOL, rtol=POL_RTOL)
        np.testing.assert_allclose(v1[0], vo[i], atol=VAL_ATOL)
        worst = max(worst, float(np.abs(p1[0] - po[i]).max()))
    for nb in (8, 16):
        idx = (pick * 6)[:nb]
        pb, vb = net.infer(x[idx])
        for r, b in enumerate(idx):
            i = pick.index(b)
            np.testing.assert_allclose(pb[r], po[i], atol=POL_ATOL, rtol=POL_RTOL)
            np.testing.assert_allclose(vb[r], vo[i], atol=VAL_ATOL)
    print("\n[configs[4] parity] L=40 batch 1/8/16 vs oracle: max|dpolicy| = %.3g" % worst)
    net.close()


@pytest.mark.parametrize("lanes,latency,forced", [(1, True, None), (8, True, None), (16, False, None), (16, True, "wino_h2"), (8, True, "wino_h2")])
def test_config4_one_tree_search_bit_exact_vs_oracle(ctx, lanes, latency, forced):
    """(iii-b) one tree, 40 blocks, sequential and lane rounds of 8 / 16, from a mid-game position through the single-tree
    boundary (agz_mcts_*): device tree == oracle tree given the same network outputs.  Batches 1 and 8 both take the latency
    regime (bit-identical per board); a 16-lane round leaves it, so that case pins one regime for every batch size.  forced:
    the tournament setting bench.py's latency leg measures — AGZ_COMPUTE_WINO_H2 | AGZ_COMPUTE_FORCE keeps the Winograd fp16x2
    tower at every batch size (lane rounds of 16 boards and the batch-1 prepareRoot run the same arithmetic per board)."""
    L, budget = 40, 48
    net = std_net(ctx, L)
    net.set_latency_mode(latency)
    if forced:
        net.set_compute_mode(MODES[forced] | capi.COMPUTE_FORCE)
        x, _ = cached_midgame(512)
        p1, v1 = net.infer(x[3:4])
        p16, v16 = net.infer(x[:16])
        np.testing.assert_array_equal(p16[3], p1[0])        # one arithmetic whatever the batch size
        np.testing.assert_array_equal(v16[3], v1[0])
    g = O.Game(O.WQ, S, S, 0, 7.5)
    rng = np.random.default_rng(11)
    player, moves, boards = O.BLACK, [], []
    g.set_to_move(player)
    for _ in range(90):
        empt = np.where(g.board() == 0)[0]
        mv = next(int(c) for c in rng.permutation(empt) if g.check(player, int(c)))
        g.apply(player, mv)
        moves.append(mv)
        boards.append(g.board())
        player = O.WHITE if player == O.BLACK else O.BLACK
        g.set_to_move(player)
    dev = A.Mcts(ctx, capi.GAME_WQ, S, S, 0, 7.5, encoder=capi.ENC_WQ, Budget=budget)
    dev.set_inferencer(capi.INF_NET, net)
    dev.set_parallel(lanes)
    orc = O.Mcts(g, enc=O.ENC_WQ, Budget=budget, lanes=lanes)

    def cb(planes):
        p, v = net.infer(planes.reshape(1, F, S, S))
        return p[0], float(v[0])

    orc.set_callback(cb, ASPACE)
    for turn in range(2):
        dev.set_game(board=g.board(), to_move=player, n_moves=len(moves), passes=0, hash=g.hash(), last_moves=moves,
                     historical=np.array(boards[-8:], np.int32))
        orc.set_game(g)
        bd, bo = dev.search(player), orc.search(player)
        assert bd == bo
        omv, ovis, obs, _ = orc.root_children()
        dmv, dvis, dbs, _ = dev.root_children()
        np.testing.assert_array_equal(dmv, omv)
        np.testing.assert_array_equal(dvis, ovis)
        np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32))
        g.apply(player, bd)
        moves.append(bd)
        boards.append(g.board())
        player = O.WHITE if player == O.BLACK else O.BLACK
        g.set_to_move(player)
    dev.close()
    net.close()


# ---- the search pinned at the depth bench.py measures -------------------------------------------------------------

def _replay_opening(o, seed, g, n):
    for _ in range(int(n)):
        o.random_move(seed, g)


def test_headline_budget_800_three_plies_on_the_measured_network_bit_exact(ctx):
    """1(a): BASELINE configs[3] per GPU AT ITS OWN BUDGET — 19x19, K=256, 20 blocks, 512 games, AGZ_COMPUTE_WINO_H2, **800 simulations per
    move**, the node pools bench.py runs with (max_nodes = 0: the library's default from the Budget), three searched plies: the third ply
    searches a tree RE-ROOTED from an 800-visit tree through the packed prepareRoot batch (mcts/search.go:92-164, 424-500).  One watched
    game against an oracle arena fed the same network (agz_net_min_same_batch copies of the leaf: the 512-board batch's kernels) —
    children, visit counts, blackScores bits, prior bits, the move — and no pool of any of the 1024 trees may overflow."""
    L, G, budget, seed, g = 20, 512, 800, 4242, 137
    net = std_net(ctx, L)
    net.set_compute_mode(MODES["wino_h2"])
    dev = A.Arena(ctx, capi.GAME_WQ, S, S, 0, 7.5, encoder=capi.ENC_WQ, n_games=G, seed=seed, Budget=budget)
    dev.set_inferencer(0, capi.INF_NET, net)
    dev.set_inferencer(1, capi.INF_NET, net)
    ab = np.array([(i % 2) == 0 for i in range(G)], dtype=np.uint8)
    dev.reset(ab)
    rng = np.random.default_rng(seed)
    n_moves = rng.integers(0, 217, size=G).astype(np.int32)
    n_moves[g] = 120
    dev.random_moves(n_moves, seed)
    nb = net.min_same_batch(1, G)

    def cb(planes):
        p, v = net.infer(np.repeat(planes.reshape(1, F, S, S), nb, axis=0))
        return p[0], float(v[0])

    # (soak knobs: AGZ_HEADLINE_NET_PLIES plies, AGZ_HEADLINE_NET_WATCH further watched games besides game 137 — profiles/r06/headline_depth_soak.txt)
    plies = int(os.environ.get("AGZ_HEADLINE_NET_PLIES", "3"))
    extra = int(os.environ.get("AGZ_HEADLINE_NET_WATCH", "0"))
    watch = [g] + [w for w in range(5, G, max(1, G // max(extra, 1)))if w != g][:extra]
    orcs = {}
    for w in watch:
        o = O.Arena(O.WQ, S, S, 0, 7.5, enc=O.ENC_WQ, Budget=budget, seed=seed + w)
        o.set_callback(0, cb, ASPACE)
        o.set_callback(1, cb, ASPACE)
        o.begin(int(ab[w]))
        _replay_opening(o, seed, w, n_moves[w])
        np.testing.assert_array_equal(dev.history(w), o.history())
        orcs[w] = o
    root_visits, nodes, preps = [], [], []
    for ply in range(plies):
        dev.begin_move()
        preps.append(dev.last_prep_batch())
        dev.simulate(budget)
        dev.end_move(True)
        for w, o in orcs.items():
            _, st0 = o.state()
            agent = 0 if ((st0["to_move"] == O.BLACK) == bool(ab[w])) else 1
            o.step(True)
            omv, ovis, obs, opr = o.root_children(agent)
            dmv, dvis, dbs, dpr = dev.root_children(w, agent)
            np.testing.assert_array_equal(dmv, omv, err_msg="game %d ply %d" % (w, ply))
            np.testing.assert_array_equal(dvis, ovis, err_msg="game %d ply %d" % (w, ply))
            np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32))
            np.testing.assert_array_equal(dpr.view(np.uint32), opr.view(np.uint32))
            assert dev.history(w)[-1] == o.history()[-1]
            if w == g:
                root_visits.append(int(ovis.sum()))
                nodes.append(dev.tree_nodes(g, agent))
        if plies > 3:
            print("[headline depth, network] ply %d: %d watched games equal so far" % (ply + 1, len(orcs)), flush=True)
    st = dev.stats()
    assert st["tree_full"] == 0 and (st["sims_total"] == plies * G * budget if plies == 3 else st["sims_total"] <= plies * G * budget)
    assert root_visits[0] >= budget and root_visits[2] > root_visits[0], root_visits   # ply 3: this search's 800 on top of the kept subtree
    assert min(nodes) > 100000, nodes
    assert preps[0] == (G, G) and 0 < preps[2][1] < G and preps[2][0] <= G, preps     # the re-rooted ply: only the roots without children
    print("\n[headline depth] budget 800 x %d plies, %d watched game(s): root visits %r, tree nodes %r, prepareRoot batches %r" % (plies, len(watch), root_visits, nodes, preps))
    dev.close()
    net.close()


def test_headline_budget_800_hash_inferencer_24_watched_games_bit_exact(ctx):
    """1(b): the same arena shape and Budget (19x19, 512 games, 800 simulations per move, three plies, default pools) with AGZ_INF_HASH — the
    oracle's cost is then MCTS only — and 24 watched games spread over the opening depths (0 .. 216 moves) and both colours: every watched
    root after every ply (children, visits, blackScores bits, prior bits), every move, and the finished examples of the watched games."""
    G, budget, seed = 512, 800, 99
    # (soak knobs: AGZ_HEADLINE_PLIES / AGZ_HEADLINE_WATCH run the same comparison longer and wider — profiles/r06/headline_depth_soak.txt)
    plies, n_watch = int(os.environ.get("AGZ_HEADLINE_PLIES", "3")), int(os.environ.get("AGZ_HEADLINE_WATCH", "24"))
    dev = A.Arena(ctx, capi.GAME_WQ, S, S, 0, 7.5, encoder=capi.ENC_WQ, n_games=G, seed=seed, Budget=budget)
    dev.set_inferencer(0, capi.INF_HASH)
    dev.set_inferencer(1, capi.INF_HASH)
    ab = np.array([(i % 3) != 0 for i in range(G)], dtype=np.uint8)
    dev.reset(ab)
    rng = np.random.default_rng(seed)
    n_moves = rng.integers(0, 217, size=G).astype(np.int32)
    order = np.argsort(n_moves, kind="stable")
    watch = sorted(set(int(order[i]) for i in np.linspace(0, G - 1, n_watch).round().astype(int)))
    assert len(watch) == n_watch
    dev.random_moves(n_moves, seed)
    orcs = {}
    for g in watch:
        o = O.Arena(O.WQ, S, S, 0, 7.5, enc=O.ENC_WQ, Budget=budget, seed=seed + g)
        o.set_inferencer(0, O.INF_HASH)
        o.set_inferencer(1, O.INF_HASH)
        o.begin(int(ab[g]))
        _replay_opening(o, seed, g, n_moves[g])
        np.testing.assert_array_equal(dev.history(g), o.history())
        orcs[g] = o
    from concurrent.futures import ThreadPoolExecutor
    for ply in range(plies):
        dev.begin_move()
        dev.simulate(budget)
        dev.end_move(True)
        agents = {}
        for g, o in orcs.items():
            _, st0 = o.state()
            agents[g] = 0 if ((st0["to_move"] == O.BLACK) == bool(ab[g])) else 1
        with ThreadPoolExecutor(max_workers=min(int(os.environ.get("AGZ_HEADLINE_THREADS", "8")), os.cpu_count() or 2)) as ex:      # (the oracle calls run outside the GIL)
            list(ex.map(lambda o: o.step(True), orcs.values()))
        for g, o in orcs.items():
            omv, ovis, obs, opr = o.root_children(agents[g])
            dmv, dvis, dbs, dpr = dev.root_children(g, agents[g])
            np.testing.assert_array_equal(dmv, omv, err_msg="game %d ply %d" % (g, ply))
            np.testing.assert_array_equal(dvis, ovis, err_msg="game %d ply %d" % (g, ply))
            np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32))
            np.testing.assert_array_equal(dpr.view(np.uint32), opr.view(np.uint32))
            assert dev.history(g)[-1] == o.history()[-1]
            if ply == 2:
                assert int(ovis.sum()) > budget + len(ovis)          # the re-rooted tree kept visits from ply 0
        if plies > 3 and ply % 8 == 7:
            print("[headline depth, hash] ply %d: %d watched games equal so far" % (ply + 1, len(orcs)), flush=True)
    st = dev.stats()
    assert st["tree_full"] == 0 and (st["sims_total"] == plies * G * budget if plies == 3 else st["sims_total"] <= plies * G * budget)   # (a long soak sees games end)
    print("\n[headline depth, hash] %d plies x %d watched games of %d at Budget %d: bit-exact; nodes per pool now %r" % (plies, n_watch, G, budget, dev.pool_capacity()))
    dev.close()


@pytest.mark.parametrize("lanes,forced", [(1, None), (8, None), (16, "wino_h2")])
def test_config4_budget_1600_one_tree_bit_exact_vs_oracle(ctx, lanes, forced):
    """1(c): BASELINE configs[4] AT ITS OWN BUDGET — one tree, 40 blocks, **1600 simulations per move**, the three operating points
    bench.py's latency leg reports (sequential and lane rounds of 8 on the latency kernel, lane rounds of 16 on the forced Winograd tower),
    two turns from a 90-move position through the single-tree boundary, the second turn re-rooting the first's tree: device tree ==
    oracle tree given the same network outputs (children, visits, blackScores bits, priors bits, move)."""
    L, budget = 40, 1600
    net = std_net(ctx, L)
    net.set_latency_mode(True)
    if forced:
        net.set_compute_mode(MODES[forced] | capi.COMPUTE_FORCE)
    g = O.Game(O.WQ, S, S, 0, 7.5)
    rng = np.random.default_rng(23)
    player, moves, boards = O.BLACK, [], []
    g.set_to_move(player)
    for _ in range(90):
        empt = np.where(g.board() == 0)[0]
        mv = next(int(c) for c in rng.permutation(empt) if g.check(player, int(c)))
        g.apply(player, mv)
        moves.append(mv)
        boards.append(g.board())
        player = O.WHITE if player == O.BLACK else O.BLACK
        g.set_to_move(player)
    dev = A.Mcts(ctx, capi.GAME_WQ, S, S, 0, 7.5, encoder=capi.ENC_WQ, Budget=budget)
    dev.set_inferencer(capi.INF_NET, net)
    dev.set_parallel(lanes)
    orc = O.Mcts(g, enc=O.ENC_WQ, Budget=budget, lanes=lanes)

    def cb(planes):
        p, v = net.infer(planes.reshape(1, F, S, S))
        return p[0], float(v[0])

    orc.set_callback(cb, ASPACE)
    vis_sum = []
    turns = int(os.environ.get("AGZ_CONFIG4_TURNS", "2"))     # (soak knob: more turns on the same handle — profiles/r06/headline_depth_soak.txt)
    for turn in range(turns):
        dev.set_game(board=g.board(), to_move=player, n_moves=len(moves), passes=0, hash=g.hash(), last_moves=moves,
                     historical=np.array(boards[-8:], np.int32))
        orc.set_game(g)
        bd, bo = dev.search(player), orc.search(player)
        assert bd == bo
        omv, ovis, obs, opr = orc.root_children()
        dmv, dvis, dbs, dpr = dev.root_children()
        np.testing.assert_array_equal(dmv, omv)
        np.testing.assert_array_equal(dvis, ovis)
        np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32))
        np.testing.assert_array_equal(dpr.view(np.uint32), opr.view(np.uint32))
        vis_sum.append(int(ovis.sum()))
        g.apply(player, bd)           # the next turn searches for the other colour: its root is this search's most visited child
        moves.append(bd)
        boards.append(g.board())
        player = O.WHITE if player == O.BLACK else O.BLACK
        g.set_to_move(player)
    assert vis_sum[1] > budget + 362      # turn 2 searched a re-rooted tree (the kept subtree's visits on top of its own 1600)
    assert dev.stats()["tree_full"] == 0
    assert dev.nodes() > 100000
    if turns > 2:
        print("\n[configs[4] depth] lanes %d: %d turns at Budget %d bit-exact; root visits %r" % (lanes, turns, budget, vis_sum))
    dev.close()
    net.close()
