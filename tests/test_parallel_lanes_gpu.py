"""GPU parity: lane-parallel search (agz_arena_set_parallel) — rounds of V simulations per tree whose leaves are one
batch, with the reference's stored virtual loss between the lanes of a round.  The oracle restates the same
deterministic semantics (oracle/mcts.hpp MCTS::parallelRound); the device must match it bit for bit: trees, moves,
boards, examples.  V = 1 is the sequential search and is covered by every other engine test."""
import numpy as np
import pytest

import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi
from test_engine_gpu import run_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("V", [2, 3, 8])
def test_tictactoe_lanes(ctx, V):
    run_pair(ctx, capi.GAME_MNK, 3, 3, 3, budget=40, a_is_black=(1, 0), parallel=V)


@pytest.mark.parametrize("V,budget", [(4, 30), (8, 30), (8, 7), (16, 50)])
def test_wq_lanes_including_partial_last_round(ctx, V, budget):
    """budget not a multiple of V: the last round runs budget % V lanes"""
    run_pair(ctx, capi.GAME_WQ, 5, 5, komi=0.5, enc=capi.ENC_WQ, budget=budget, a_is_black=(1, 0), max_moves=40, parallel=V)


def test_connect4_and_komi_lanes(ctx):
    run_pair(ctx, capi.GAME_C4, 6, 7, 4, budget=36, a_is_black=(1, 0), parallel=4, n_plies=14)
    run_pair(ctx, capi.GAME_KOMI, 5, 5, 3, budget=36, a_is_black=(1,), parallel=4)


def test_lanes_differ_from_sequential_but_conserve_visits(ctx):
    """the lane search is a different (not a worse) set of simulations: visits are conserved, the tree differs"""
    res = {}
    for V in (1, 8):
        dev = A.Arena(ctx, capi.GAME_WQ, 9, 9, 0, 7.5, encoder=capi.ENC_WQ, n_games=1, Budget=64)
        dev.set_inferencer(0, capi.INF_HASH)
        dev.set_inferencer(1, capi.INF_HASH)
        dev.set_parallel(V)
        dev.reset(np.array([1], dtype=np.uint8))
        dev.begin_move()
        dev.simulate(64)
        dev.end_move(False)
        mv, vis, bs, pr = dev.root_children(0, 0)
        assert int(vis.sum()) - len(vis) == 64
        res[V] = (mv.copy(), vis.copy(), dev.tree_nodes(0, 0))
        st = dev.stats()
        assert st["sims_total"] == 64
    # Black to move at the root: the stored virtual loss only enters White's Evaluate (node.go:150-152), so the lanes of a
    # round share the first move and fan out one level down; lanes arriving at a leaf whose expansion is in flight back
    # the same value up again instead of expanding deeper -> fewer nodes than the sequential search
    assert res[8][2] < res[1][2], (res[1][2], res[8][2])


def test_lanes_with_the_network_batch(ctx):
    """NET inferencer: a round of V lanes is ONE batch of V rows per game (lane-major slots); oracle fed through a callback
    evaluating with the same regime (batch of V copies)."""
    S, K, L, F, V, budget = 5, 64, 1, 18, 4, 24
    net = A.Net(ctx, K, L, 32, S, S, F, S * S + 1, bn_mode=capi.BN_IDENTITY)
    net.init_random(3)
    for i in range(net.num_params()):
        nm, n = net.param_info(i)
        if nm.endswith("_gamma"):
            net.set_param(i, np.ones(n, np.float32))
        elif nm.endswith("_beta"):
            net.set_param(i, np.zeros(n, np.float32))
    net.commit()
    net.set_latency_mode(False)    # one arithmetic regime whatever the batch size: bitwise batch independence
    dev = A.Arena(ctx, capi.GAME_WQ, S, S, 0, 0.5, encoder=capi.ENC_WQ, n_games=2, Budget=budget, max_moves=30)
    dev.set_inferencer(0, capi.INF_NET, net)
    dev.set_inferencer(1, capi.INF_NET, net)
    dev.set_parallel(V)
    ab = np.array([1, 0], dtype=np.uint8)
    dev.reset(ab)

    def cb(planes):
        p, v = net.infer(planes.reshape(1, F, S, S))
        return p[0], float(v[0])

    orcs = []
    for g in range(2):
        o = O.Arena(O.WQ, S, S, 0, 0.5, enc=O.ENC_WQ, Budget=budget, max_moves=30)
        o.set_callback(0, cb, S * S + 1)
        o.set_callback(1, cb, S * S + 1)
        o.set_parallel(V)
        o.begin(int(ab[g]))
        orcs.append(o)
    for ply in range(6):
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
            assert dev.history(g)[-1] == o.history()[-1]


def test_set_parallel_argument_checks(ctx):
    dev = A.Arena(ctx, capi.GAME_MNK, 3, 3, 3, n_games=1, Budget=4)
    for bad in (0, 17, -3):
        with pytest.raises(A.AgzError, match="lanes"):
            dev.set_parallel(bad)
    dev.set_inferencer(0, capi.INF_HASH)
    dev.set_inferencer(1, capi.INF_HASH)
    dev.reset(np.array([1], dtype=np.uint8))
    dev.begin_move()
    with pytest.raises(A.AgzError, match="in progress"):
        dev.set_parallel(2)
