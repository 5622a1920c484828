"""GPU: synthetic openings (agz_arena_random_moves; SURVEY 8(d) "u uniformly-random legal moves from the empty board") are
bit-exact against the oracle's restatement (orc_arena_random_move): same moves, same boards, and searches started from those
positions stay bit-exact — this is how bench.py gives every one of its 512 games its own mid-game position."""
import numpy as np
import pytest

import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind,okind,m,n,k,komi,enc,G,most", [
    (capi.GAME_MNK, O.MNK, 5, 5, 4, 0.0, 0, 6, 12),
    (capi.GAME_C4, O.C4, 6, 7, 4, 0.0, 0, 6, 20),
    (capi.GAME_KOMI, O.KOMI, 5, 5, 3, 0.0, 0, 6, 14),
    (capi.GAME_WQ, O.WQ, 9, 9, 0, 7.5, 1, 8, 60),
    (capi.GAME_WQ, O.WQ, 19, 19, 0, 7.5, 1, 8, 216),
])
def test_random_openings_match_the_oracle_and_searches_continue_bit_exact(ctx, kind, okind, m, n, k, komi, enc, G, most):
    budget, seed = 24, 1337
    rng = np.random.default_rng(G * 31 + most)
    n_moves = rng.integers(0, most + 1, size=G).astype(np.int32)
    n_moves[0], n_moves[1] = most, 0
    dev = A.Arena(ctx, kind, m, n, k, komi, encoder=enc, n_games=G, Budget=budget)
    dev.set_inferencer(0, capi.INF_HASH)
    dev.set_inferencer(1, capi.INF_HASH)
    ab = np.array([(g % 3) != 0 for g in range(G)], dtype=np.uint8)
    dev.reset(ab)
    dev.random_moves(n_moves, seed)
    orcs = []
    for g in range(G):
        o = O.Arena(okind, m, n, k, komi, enc=enc, Budget=budget)
        o.set_inferencer(0, O.INF_HASH)
        o.set_inferencer(1, O.INF_HASH)
        o.begin(int(ab[g]))
        for _ in range(int(n_moves[g])):
            if o.random_move(seed, g) == -32768:
                break
        orcs.append(o)
        np.testing.assert_array_equal(dev.history(g), o.history(), err_msg="game %d" % g)
        db, dst = dev.game(g)
        ob, ost = o.state()
        np.testing.assert_array_equal(db, ob, err_msg="game %d" % g)
        assert (dst["ended"], dst["winner"]) == (ost["ended"], ost["winner"]), g
        if not ost["ended"] and kind != capi.GAME_C4:   # (c4's Apply never flips nextToMove, game/c4/game.go:56-73: the oracle's
            assert dst["to_move"] == ost["to_move"], g   #  game reports a stale colour until the next Search sets it)
    # distinct positions, then two searched plies from them (trees, moves, boards)
    if kind == capi.GAME_WQ:
        assert len({dev.game(g)[0].tobytes() for g in range(G)}) >= G - 1
    for ply in range(2):
        dev.begin_move()
        dev.simulate(budget)
        dev.end_move(True)
        for g, o in enumerate(orcs):
            _, st0 = o.state()
            if st0["ended"]:
                continue
            mover = O.BLACK if len(o.history()) % 2 == 0 else O.WHITE   # colours alternate from Black (no ignored passes here)
            agent = 0 if ((mover == O.BLACK) == bool(ab[g])) else 1
            o.step(True)
            omv, ovis, obs, _ = o.root_children(agent)
            dmv, dvis, dbs, _ = dev.root_children(g, agent)
            np.testing.assert_array_equal(dmv, omv, err_msg="game %d ply %d" % (g, ply))
            np.testing.assert_array_equal(dvis, ovis, err_msg="game %d ply %d" % (g, ply))
            np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32))
            assert dev.history(g)[-1] == o.history()[-1]
    # examples recorded from those positions carry the history planes of the random prefix
    dp, dq, dv, dg = dev.examples()
    for g, o in enumerate(orcs):
        op, oq, ov = o.examples()
        rows = np.where(dg == g)[0]
        assert len(rows) == len(ov)
        for r, i in zip(rows, range(len(ov))):
            np.testing.assert_array_equal(dp[r].view(np.uint32), op[i].view(np.uint32), err_msg="game %d example %d" % (g, i))
            np.testing.assert_array_equal(dq[r], oq[i])


def test_two_nets_with_games_at_different_plies(ctx):
    """the A/B sub-batch split followed a host-side ply parity that agz_arena_apply_moves flipped even for
    skipped games.  The split now reads who is to move from the device: games at different plies (here after random openings of
    different lengths and an AGZ_NO_MOVE) search with the right net — checked against the oracle with two different nets."""
    S, K, L, F, G, budget = 5, 32, 1, 18, 4, 16
    nets = []
    for sd in (3, 4):
        net = A.Net(ctx, K, L, 2 * K, S, S, F, S * S + 1, bn_mode=capi.BN_IDENTITY)
        net.init_random(sd)
        for i in range(net.num_params()):
            name, cnt = net.param_info(i)
            if name.endswith("_gamma"):
                net.set_param(i, np.ones(cnt, np.float32))
            elif name.endswith("_beta"):
                net.set_param(i, np.zeros(cnt, np.float32))
        net.commit()
        net.set_latency_mode(False)
        nets.append(net)
    dev = A.Arena(ctx, capi.GAME_WQ, S, S, 0, 0.5, encoder=capi.ENC_WQ, n_games=G, Budget=budget)
    dev.set_inferencer(0, capi.INF_NET, nets[0])
    dev.set_inferencer(1, capi.INF_NET, nets[1])
    ab = np.array([1, 0, 1, 0], dtype=np.uint8)
    dev.reset(ab)
    n_moves = np.array([0, 1, 2, 3], np.int32)    # mixed plies: A moves in games 0 and 1... by colour, not by a global parity
    dev.random_moves(n_moves, 99)
    mv = np.full(G, capi.NO_MOVE, np.int32)
    mv[2] = int(np.where(dev.game(2)[0] == 0)[0][0])   # one more ply for game 2 only
    dev.apply_moves(mv)

    def mk(net):
        def cb(planes):
            p, v = net.infer(planes.reshape(1, F, S, S))
            return p[0], float(v[0])
        return cb

    orcs = []
    for g in range(G):
        o = O.Arena(O.WQ, S, S, 0, 0.5, enc=O.ENC_WQ, Budget=budget)
        o.set_callback(0, mk(nets[0]), S * S + 1)
        o.set_callback(1, mk(nets[1]), S * S + 1)
        o.begin(int(ab[g]))
        for _ in range(int(n_moves[g])):
            o.random_move(99, g)
        if g == 2:
            assert o.apply_move(int(mv[2])) == 1
        orcs.append(o)
    for ply in range(3):
        dev.begin_move()
        dev.simulate(budget)
        dev.end_move(True)
        for g, o in enumerate(orcs):
            _, st0 = o.state()
            if st0["ended"]:
                continue
            agent = 0 if ((st0["to_move"] == O.BLACK) == bool(ab[g])) else 1
            o.step(True)
            omv, ovis, obs, _ = o.root_children(agent)
            dmv, dvis, dbs, _ = dev.root_children(g, agent)
            np.testing.assert_array_equal(dmv, omv, err_msg="game %d ply %d" % (g, ply))
            np.testing.assert_array_equal(dvis, ovis, err_msg="game %d ply %d" % (g, ply))
            np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32))
    with pytest.raises(A.AgzError, match="one lane"):
        dev2 = A.Arena(ctx, capi.GAME_WQ, S, S, 0, 0.5, encoder=capi.ENC_WQ, n_games=2, Budget=4)
        dev2.set_inferencer(0, capi.INF_NET, nets[0])
        dev2.set_parallel(4)
        dev2.set_inferencer(1, capi.INF_NET, nets[1])   # lanes first, then a second net


def test_unlabelled_examples_stay_in_the_arena(ctx):
    """agz_examples_append_arena only takes examples of finished games; a full buffer is counted."""
    G, budget = 8, 6
    dev = A.Arena(ctx, capi.GAME_MNK, 3, 3, 3, n_games=G, Budget=budget)
    dev.set_inferencer(0, capi.INF_HASH)
    dev.set_inferencer(1, capi.INF_HASH)
    dev.reset()
    dev.play(3, record=True)     # three plies: no tic-tac-toe game can be over
    ex = A.Examples(ctx, 2, 3, 3, 10)
    ex.append_arena(dev)
    assert len(ex) == 0 and dev.stats()["examples"] == 3 * G
    dev.play(0, record=True)     # to the end
    ex.append_arena(dev)
    _, _, v = ex.get()
    assert len(ex) == dev.stats()["examples"] and set(np.unique(v)).issubset({-1.0, 0.0, 1.0})
    assert dev.stats()["examples_dropped"] == 0
