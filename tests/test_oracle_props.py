"""CPU: properties of the oracle's restatement (host logic the GPU engine must reproduce) — encoders, rule
quirks, search invariants.  Small, fast; the bit-exact GPU-vs-oracle comparisons live in test_engine_gpu.py."""
import numpy as np
import pytest

import oracle_lib as O


def test_twoplane_encoder():  # cmd/tictactoe/main.go:26-47
    g = O.Game(O.MNK, 3, 3, 3)
    g.set_board([1, 2, 0, 0, 0, 0, 0, 0, 0])
    g.set_to_move(O.WHITE)
    e = g.encode(O.ENC_TWOPLANE)
    np.testing.assert_allclose(e[:9], [1, -1] + [0.001] * 7)
    np.testing.assert_allclose(e[9:], [-1] * 9)


def test_wq_encoder_never_encodes_current_board_and_uses_negative_zero():  # encoding_helper.go:29-68
    g = O.Game(O.WQ, 5, 5, komi=0.5)
    g.set_to_move(O.BLACK)
    moves = [(O.BLACK, 0), (O.WHITE, 6), (O.BLACK, 12), (O.WHITE, 18), (O.BLACK, 24)]
    for p, m in moves:
        g.apply(p, m)
    e = g.encode(O.ENC_WQ).reshape(18, 25)
    assert g.to_move() == O.WHITE
    # White to move: "black" half is planes 8..15, "white" half 0..7, plane 17 = -1, plane 16 = 0
    assert np.all(e[17] == -1) and np.all(e[16] == 0)
    # slot 0 = board after move 4 (h = current-1 = 3): stones 0,12 black; 6,18 white ; the 5th stone (24) absent
    assert e[8][0] == 1 and e[8][12] == 1 and e[8][6] == -1 and e[8][18] == -1 and e[8][24] == 0
    assert e[0][0] == -1 and e[0][6] == 1
    assert np.signbit(e[0][24]) and e[0][24] == 0  # vecf32.Scale(-1) turns empty cells into -0.0
    assert np.all(e[7] == 0) and np.all(e[15] == 0)  # slot 7 of each half is never written
    # h > 0 excludes the board after move 1
    assert np.all(e[8 + 3] == 0) and not np.any(np.signbit(e[3]))


def test_reference_suicide_rule_ignores_friendly_liberties():
    """check() runs before the stone is placed (komi/game.go:340-344): a point with no empty neighbour is illegal
    unless it captures, even when it connects to a friendly group that has liberties."""
    g = O.Game(O.KOMI, 3, 3, 3)
    g.set_board([0, 1, 0,
                 1, 0, 1,
                 0, 1, 0])
    assert not g.check(O.WHITE, 4)   # plain suicide
    assert not g.check(O.BLACK, 4)   # connects four live black stones — still rejected by the reference rule
    assert not g.check(O.BLACK, 0)   # corner between two black stones: no empty neighbour either
    g2 = O.Game(O.KOMI, 3, 3, 3)
    g2.set_board([0, 1, 0, 0, 0, 0, 0, 0, 0])
    assert g2.check(O.BLACK, 0) and g2.check(O.WHITE, 0)  # an empty neighbour (point 3) makes it legal


def test_duplicate_capture_counting():  # a dead group touched on two sides is listed twice
    g = O.Game(O.KOMI, 3, 3, 9)
    # white L-group {1, 4}... build: black plays at 0 touching white group {1,3}? use a bent group around the corner
    g.set_board([0, 2, 1,
                 2, 2, 1,
                 1, 1, 0])
    taken = g.komi_apply(O.BLACK, 0)  # corner point touches white stones 1 and 3, both in the same dead group {1,3,4}
    assert taken == 6  # 3 stones listed twice
    np.testing.assert_array_equal(g.board(), [1, 0, 1, 0, 0, 1, 1, 1, 0])


def test_mnk_hash_is_fnv1a_of_colour_names():  # mnk.go:76-82
    g = O.Game(O.MNK, 1, 1, 1)
    h = 2166136261
    for ch in b"None":
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    assert g.hash() == h


def test_c4_apply_never_flips_to_move():
    g = O.Game(O.C4, 6, 7, 4)
    g.set_to_move(O.BLACK)
    g.apply(O.BLACK, 3)
    assert g.to_move() == O.BLACK and g.move_number() == 1
    assert g.board()[5 * 7 + 3] == O.BLACK  # dropped to the bottom row
    assert g.check(O.BLACK, O.PASS)


def test_wq_completion_pass_and_area_score():
    g = O.Game(O.WQ, 3, 3, komi=0.5)
    g.set_board([1, 1, 0,
                 1, 1, 2,
                 0, 2, 2])
    assert g.score(O.BLACK) == 4 and g.score(O.WHITE) == 3  # the two empty points touch both colours
    g.apply(O.BLACK, O.PASS)
    assert g.passes() == 1 and not g.ended()[0]
    g.apply(O.WHITE, O.PASS)
    ended, winner = g.ended()
    assert ended and winner == O.BLACK  # 4 vs 3 + 0.5


@pytest.mark.parametrize("budget", [1, 10, 57])
def test_search_invariants(budget):
    """root visits = 1 (creation) + 1 (prepareRoot) + non-null sims; children visits - 1 sum to the sims that
    passed through them (tree.go:110, search.go:392-408, node.go:70-76)."""
    a = O.Arena(O.MNK, 3, 3, 3, Budget=budget)
    a.set_inferencer(0, O.INF_HASH)
    a.set_inferencer(1, O.INF_HASH)
    a.begin(1)
    a.step()
    st = a.tree_stats(0)
    mv, vis, bs, pr = a.root_children(0)
    assert st["root_visits"] == 2 + st["playouts"]
    assert int(vis.sum()) - len(vis) == st["playouts"]
    assert abs(float(pr.sum()) - 1.0) < 1e-5
    assert list(vis) == sorted(vis, reverse=True)  # fancySort: visits descending
    assert st["nn_evals"] == 1 + st["playouts"]


def test_policies_are_one_hot_on_the_chosen_move():
    a = O.Arena(O.MNK, 3, 3, 3, Budget=20)
    a.set_inferencer(0, O.INF_HASH)
    a.set_inferencer(1, O.INF_HASH)
    a.begin(1)
    a.play(0, True)
    B, P, V = a.examples()
    hist = a.history()
    assert P.shape == (len(hist), 10)
    for i, m in enumerate(hist):
        assert P[i, m] == 1.0 and P[i].sum() == 1.0
    _, st = a.state()
    assert st["ended"] == 1
    assert set(np.unique(V)).issubset({-1.0, 0.0, 1.0})


@pytest.mark.parametrize("lanes", [2, 4, 8])
def test_lane_parallel_rounds_conserve_visits_and_are_deterministic(lanes):
    """oracle/mcts.hpp MCTS::parallelRound (MCTSConfig::Parallel): every lane of a round backs exactly one value up (or
    none, for a null lane), so the root's child visits still sum to Budget + #children; same inputs, same tree."""
    def one():
        a = O.Arena(O.WQ, 5, 5, komi=0.5, enc=O.ENC_WQ, Budget=37)      # 37: the last round is partial
        a.set_inferencer(0, O.INF_HASH)
        a.set_inferencer(1, O.INF_HASH)
        a.set_parallel(lanes)
        a.begin(1)
        a.step(True)
        return a.root_children(0)
    mv, vis, bs, pr = one()
    assert int(vis.sum()) - len(vis) == 37
    mv2, vis2, bs2, _ = one()
    assert np.array_equal(mv, mv2) and np.array_equal(vis, vis2) and np.array_equal(bs.view(np.uint32), bs2.view(np.uint32))


def test_lane_parallel_virtual_loss_only_steers_white():
    """node.go:147-159: the stored virtual loss enters Evaluate for White only.  With Black to move at the root all lanes
    of a round share the first move; with White to move they fan out over the root's children."""
    def first_round_spread(a_is_black):
        # one round of 8 lanes on a fresh tree: budget 8
        a = O.Arena(O.MNK, 5, 5, 4, Budget=8)
        a.set_inferencer(0, O.INF_HASH)
        a.set_inferencer(1, O.INF_HASH)
        a.set_parallel(8)
        a.begin(a_is_black)
        a.step(True)            # Black's move (agent A or B)
        a.step(True)            # White's move: its root search is the one we look at
        agent_white = 1 if a_is_black else 0
        mv, vis, _, _ = a.root_children(agent_white)
        return int((vis > 1).sum())
    # White to move at ITS root: the 8 lanes are steered apart by the virtual loss -> several children visited
    assert first_round_spread(1) >= 3
    # Black to move at its root (first ply): no steering at the root level
    a = O.Arena(O.MNK, 5, 5, 4, Budget=8)
    a.set_inferencer(0, O.INF_HASH)
    a.set_inferencer(1, O.INF_HASH)
    a.set_parallel(8)
    a.begin(1)
    a.step(True)
    _, vis, _, _ = a.root_children(0)
    assert int((vis > 1).sum()) == 1
