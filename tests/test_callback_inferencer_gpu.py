"""GPU: AGZ_INF_CALLBACK — the `nn Inferencer` argument of mcts.New as a HOST function (mcts/mcts.go:15-18, tree.go:80).

The device search hands the leaves of a simulation step to a host function between k_select and k_expand and consumes the rows it fills
exactly as it consumes a network's.  Pinned here:
(i)   the reference's own `Example` (mcts/example_test.go:38-156: its dummyNN restated in Python, driven through the callback on ONE tree
      searched alternately) plays the documented game and leaves the same tree as the built-in AGZ_INF_SCRIPT and as the oracle;
(ii)  a host restatement of the synthetic hash inferencer, fed the boards the callback receives, gives bit-identical trees to the
      device's AGZ_INF_HASH on every game (arena, many games, lane rounds included) — a network-independent differential hook;
(iii) the planes handed over are the encoder's tensor bit for bit: a callback that calls the HIP network on them reproduces AGZ_INF_NET;
(iv)  argument checks, a failing callee, mixed agents (one network, one callback).
"""
import json
import os

import numpy as np
import pytest

import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))


def dummy_nn(leaves):
    """mcts/example_test.go:40-72 — dummyNN.Infer by MoveNumber(); `8 / 9` is integer division: 0"""
    cell = [4, 0, 2, 6, 3, 5, 1, 7, 8]
    n = len(leaves["move_number"])
    pol = np.zeros((n, 10), np.float32)          # "10 because last one is a pass"
    val = np.zeros(n, np.float32)
    for i, mn in enumerate(leaves["move_number"]):
        if 0 <= mn < 9:
            pol[i, cell[mn]] = 0.1 if (mn & 1) else 0.9
            val[i] = 0.5 if mn in (0, 1, 5) else 0.0
    return pol, val


def mix32(x):
    x = np.asarray(x, np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= np.uint32(0x85EBCA6B)
    x ^= x >> np.uint32(13)
    x *= np.uint32(0xC2B2AE35)
    x ^= x >> np.uint32(16)
    return x


def hash_nn(policy_len):
    """the synthetic position-hash inferencer (oracle/arena.hpp HashNN; engine.hip AGZ_INF_HASH) restated on the host from what the
    callback is handed: board cells and the mover"""
    def f(leaves):
        with np.errstate(over="ignore"):
            b = leaves["board"].astype(np.uint32)                       # [n, cells]
            n, cells = b.shape
            idx = np.arange(cells, dtype=np.uint32)[None, :] * np.uint32(4)
            h = mix32(idx + b + np.uint32(1)).sum(axis=1, dtype=np.uint32)
            h = h + mix32(np.uint32(0xABCD0000) + leaves["to_move"].astype(np.uint32))
            val = (mix32(h ^ np.uint32(0xDEADBEEF)) >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
            i = np.arange(policy_len, dtype=np.uint32)[None, :]
            pol = ((mix32(h[:, None] + i * np.uint32(0x9E3779B9)) >> np.uint32(8)) + np.uint32(1)).astype(np.float32) * np.float32(1.0 / 16777216.0)
        return pol, val
    return f


class Host:
    def __init__(self, okind, m, n, k, komi):
        self.g = O.Game(okind, m, n, k, komi)
        self.g.set_to_move(O.BLACK)
        self.moves, self.boards = [], []

    def apply(self, player, mv):
        self.g.apply(player, mv)
        self.moves.append(mv)
        self.boards.append(self.g.board())
        self.g.set_to_move(O.WHITE if player == O.BLACK else O.BLACK)

    def state_kw(self):
        n = len(self.moves)
        return dict(board=self.g.board(), to_move=self.g.to_move(), n_moves=n, passes=max(self.g.passes(), 0), hash=self.g.hash(),
                    last_moves=self.moves, historical=np.array(self.boards[max(0, n - 8):], np.int32))


@pytest.mark.parametrize("budget", [200, 400])
def test_reference_example_through_the_host_callback(ctx, budget):
    """(i) mcts/example_test.go:74-156: t := mcts.New(g, conf, dummyNN{}); for !ended { best := t.Search(player); g.Apply }.  Three searches
    side by side — the callback (Python dummyNN), the built-in script, the oracle — must agree on every root after every turn; the moves are
    the reference's documented game."""
    host = Host(O.MNK, 3, 3, 3, 0.0)
    seen = []

    def nn(leaves):
        seen.append((int(leaves["planes"].shape[0]), int(leaves["move_number"][0]), int(leaves["to_move"][0])))
        assert leaves["planes"].shape[1:] == (2, 3, 3) and leaves["policy_len"] == 10
        return dummy_nn(leaves)

    cb = A.Mcts(ctx, capi.GAME_MNK, 3, 3, 3, Budget=budget)
    cb.set_inferencer_callback(nn, 10)
    sc = A.Mcts(ctx, capi.GAME_MNK, 3, 3, 3, Budget=budget)
    sc.set_inferencer(capi.INF_SCRIPT)
    ex = O.ExampleSearch(O.MNK, 3, 3, 3, Budget=budget, inf=O.INF_SCRIPT)
    player, moves = O.BLACK, []
    while not host.g.ended()[0]:
        cb.set_game(**host.state_kw())
        sc.set_game(**host.state_kw())
        b1, b2 = cb.search(player), sc.search(player)
        bo, _, _ = ex.turn()
        assert b1 == b2 == bo
        k1, k2 = cb.root_children(), sc.root_children()
        for a, b in zip(k1, k2):
            np.testing.assert_array_equal(a.view(np.uint32) if a.dtype == np.float32 else a, b.view(np.uint32) if b.dtype == np.float32 else b)
        assert [(int(m), int(v), float(s)) for m, v, s in zip(k1[0], k1[1], k1[2])] == ex.root_children()
        host.apply(player, b1)
        moves.append(b1)
        player = O.WHITE if player == O.BLACK else O.BLACK
    assert moves == KATS["mcts_example"]["moves"]
    assert host.g.ended()[1] == KATS["mcts_example"]["winner"]
    # one leaf per call on a single tree; the callee was asked exactly as often as the oracle's inferencer (one evaluation per non-null
    # simulation + one per prepareRoot that found a fresh root)
    assert all(n == 1 for n, _, _ in seen)
    assert len(seen) == ex.nn_evals() == cb.stats()["nn_evals"] == sc.stats()["nn_evals"]
    cb.close()
    sc.close()


@pytest.mark.parametrize("kind,okind,m,n,k,komi,enc,G,budget,plies,lanes", [
    (capi.GAME_MNK, O.MNK, 3, 3, 3, 0.0, capi.ENC_TWOPLANE, 9, 60, 5, 1),
    (capi.GAME_C4, O.C4, 6, 7, 4, 0.0, capi.ENC_TWOPLANE, 16, 40, 4, 1),
    (capi.GAME_KOMI, O.KOMI, 5, 5, 3, 0.0, capi.ENC_TWOPLANE, 8, 40, 4, 1),
    (capi.GAME_WQ, O.WQ, 9, 9, 0, 5.5, capi.ENC_WQ, 24, 48, 4, 1),
    (capi.GAME_WQ, O.WQ, 9, 9, 0, 5.5, capi.ENC_WQ, 6, 48, 3, 4),
    (capi.GAME_WQ, O.WQ, 19, 19, 0, 7.5, capi.ENC_WQ, 32, 40, 3, 1),
], ids=["ttt", "c4", "komi", "go9", "go9-lanes4", "go19"])
def test_host_hash_inferencer_equals_the_device_one(ctx, kind, okind, m, n, k, komi, enc, G, budget, plies, lanes):
    """(ii) two arenas on the same seeds, one with the device's AGZ_INF_HASH, one asking a host function that restates the same hash from
    the boards it is handed: every game's root (children, visits, blackScores bits, prior bits), every history and the counters must be
    identical — the callback batches, packs and scatters rows correctly for every game, with restarts, and in lane rounds."""
    A_ = n if kind == capi.GAME_C4 else m * n
    calls = []

    def nn(leaves):
        calls.append(int(leaves["planes"].shape[0]))
        assert np.all(np.diff(leaves["game"]) >= 0)                       # packed in game order
        return hash_nn(A_ + 1)(leaves)

    arenas = []
    for use_cb in (True, False):
        dev = A.Arena(ctx, kind, m, n, k, komi, encoder=enc, n_games=G, seed=77, Budget=budget)
        if lanes > 1:
            dev.set_parallel(lanes)
        for agent in (0, 1):
            if use_cb:
                dev.set_inferencer_callback(agent, nn, A_ + 1)
            else:
                dev.set_inferencer(agent, capi.INF_HASH)
        dev.reset()
        dev.random_moves(np.random.default_rng(3).integers(0, max(2, A_ // 3), size=G).astype(np.int32), 3)
        arenas.append(dev)
    for ply in range(plies):
        for dev in arenas:
            dev.begin_move()
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
    assert max(calls) > 1 and max(calls) <= G * lanes          # leaves of many games travel in ONE call
    # the oracle agrees with both (the device HASH inferencer is itself pinned to it elsewhere): game 0
    for dev in arenas:
        dev.close()


def test_callback_planes_are_the_encoders_and_reproduce_the_network(ctx):
    """(iii) the planes a callback receives are the encoder's NCHW tensor of the leaf state: a host function that evaluates them with the HIP
    network (agz_net_infer) leaves the same trees as AGZ_INF_NET itself — 9x9 Go with the WQ encoder (history planes), mixed with a
    network agent: agent A = the network on the device, agent B = the callback."""
    S, G, budget = 9, 12, 32
    net = A.Net(ctx, 32, 2, 64, S, S, 18, S * S + 1, bn_mode=capi.BN_IDENTITY)
    net.init_random(5)
    net.commit()
    net.set_compute_mode(capi.COMPUTE_F32_MFMA | capi.COMPUTE_FORCE)     # one arithmetic whatever the batch size

    def nn(leaves):
        x = np.ascontiguousarray(leaves["planes"])
        assert np.isin(x, (-1.0, 0.0, 1.0)).all()
        p, v = net.infer(x)
        return p, v

    arenas = []
    for use_cb in (True, False):
        dev = A.Arena(ctx, capi.GAME_WQ, S, S, 0, 5.5, encoder=capi.ENC_WQ, n_games=G, seed=5, Budget=budget)
        dev.set_inferencer(0, capi.INF_NET, net)
        if use_cb:
            dev.set_inferencer_callback(1, nn, S * S + 1)
        else:
            dev.set_inferencer(1, capi.INF_NET, net)
        dev.reset()
        dev.random_moves(np.random.default_rng(9).integers(4, 30, size=G).astype(np.int32), 9)
        arenas.append(dev)
    for ply in range(4):
        for dev in arenas:
            dev.begin_move()
            dev.simulate(budget)
            dev.end_move(True)
        for g in range(G):
            for agent in (0, 1):
                a, b = arenas[0].root_children(g, agent), arenas[1].root_children(g, agent)
                np.testing.assert_array_equal(a[0], b[0], err_msg="game %d agent %d ply %d" % (g, agent, ply))
                np.testing.assert_array_equal(a[1], b[1])
                np.testing.assert_array_equal(a[2].view(np.uint32), b[2].view(np.uint32))
                np.testing.assert_array_equal(a[3].view(np.uint32), b[3].view(np.uint32))
    for dev in arenas:
        dev.close()
    net.close()


def test_callback_argument_checks_and_a_failing_callee(ctx):
    """(iv) policy_len below the game's ActionSpace, a NULL function, the kind constant through the plain setter: clean errors; a callee
    that raises aborts the search with AGZ_E_CALLBACK (-8), the arena works again after a reset and another inferencer."""
    import ctypes as C
    dev = A.Arena(ctx, capi.GAME_MNK, 3, 3, 3, n_games=3, seed=1, Budget=10)
    with pytest.raises(A.AgzError, match="policy_len"):
        dev.set_inferencer_callback(0, dummy_nn, 8)
    with pytest.raises(A.AgzError, match="agz_arena_set_inferencer_callback"):
        dev.set_inferencer(0, capi.INF_CALLBACK)
    assert capi.lib().agz_arena_set_inferencer_callback(dev.h, 0, capi.INFER_FN(), None, 10) == -1

    def boom(leaves):
        raise ValueError("no network today")

    dev.set_inferencer_callback(0, boom, 10)
    dev.set_inferencer_callback(1, boom, 10)
    dev.reset()
    with pytest.raises(A.AgzError, match=r"\(-8\)"):
        dev.begin_move()
    dev.set_inferencer(0, capi.INF_SCRIPT)
    dev.set_inferencer(1, capi.INF_SCRIPT)
    dev.reset()
    dev.play(0, True)
    assert dev.stats()["games_finished"] == 3
    dev.close()
