"""CPU: pins the oracle (CPU restatement) against the known-answer tables the reference's own tests hold
(tests/golden/reference_kats.json, transcribed from game/*/..._test.go, dualnet/config_test.go,
mcts/example_test.go) — SURVEY §8(c)."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "reference_kats.json")))


def test_round_table():  # dualnet/config_test.go:5-25
    for a, want in KATS["round"]:
        assert O.lib().orc_round(a) == want


@pytest.mark.parametrize("row", range(len(KATS["komi_apply"])))
def test_komi_apply(row):  # game/komi/komi_test.go:183-229
    t = KATS["komi_apply"][row]
    g = O.Game(O.KOMI, t["m"], t["n"], 3)
    g.set_board(t["board"])
    taken = g.komi_apply(t["player"], t["move"])
    if t["err"]:
        assert taken == -1
        return
    assert taken == t["taken"]
    np.testing.assert_array_equal(g.board(), t["board2"])
    assert g.score(O.WHITE) == t["white"]
    assert g.score(O.BLACK) == t["black"]


@pytest.mark.parametrize("row", range(len(KATS["wq_apply"])))
def test_wq_board_apply_and_score(row):  # game/wq/wq_test.go:198-238
    t = KATS["wq_apply"][row]
    g = O.Game(O.WQ, t["size"], t["size"])
    g.set_board(t["board"])
    taken = g.wq_board_apply(t["player"], t["move"])
    if t["err"]:
        assert taken == -1
        return
    assert taken == t["taken"]
    np.testing.assert_array_equal(g.board(), t["board2"])
    assert g.wq_board_score(O.WHITE) == t["white"]
    assert g.wq_board_score(O.BLACK) == t["black"]


def test_komi_ended():  # game/komi/komi_test.go:264-290
    t = KATS["komi_ended"]
    g = O.Game(O.KOMI, t["m"], t["n"], t["k"])
    g.set_board(t["board"])
    g.set_to_move(t["to_move"])
    assert g.ended()[0] == t["ended"]


def test_komi_check_3x7():  # game/komi/komi_test.go:292-313 (needs the reference's m-strided geometry)
    t = KATS["komi_check"]
    g = O.Game(O.KOMI, t["m"], t["n"], t["k"])
    g.set_board(t["board"])
    g.set_to_move(t["to_move"])
    assert g.check(t["player"], t["move"]) == t["legal"]


@pytest.mark.parametrize("row", range(len(KATS["mnk_winner"])))
def test_mnk_winner(row):  # game/mnk/mnk_test.go:9-73
    t = KATS["mnk_winner"][row]
    g = O.Game(O.MNK, t["m"], t["n"], t["k"])
    g.set_board(t["board"])
    assert g.is_winner(t["winner_is"])
    if t["ended"] is not None:
        assert g.ended()[0] == t["ended"]


@pytest.mark.parametrize("row", range(len(KATS["mnk_ended"])))
def test_tictactoe_ended(row):  # game/mnk/mnk_test.go:75-129
    t = KATS["mnk_ended"][row]
    g = O.Game(O.MNK, 3, 3, 3)
    g.set_board(t["board"])
    ended, winner = g.ended()
    assert ended == t["ended"] and winner == t["winner"]


@pytest.mark.parametrize("row", range(len(KATS["c4_ended"])))
def test_c4_ended(row):  # game/c4/c4_test.go:9-101
    t = KATS["c4_ended"][row]
    g = O.Game(O.C4, 6, 7, 4)
    g.set_board(t["board"])
    ended, winner = g.ended()
    assert ended == t["ended"]
    if ended:
        assert winner == t["winner"]


def test_clone_eq_reset():  # game/komi/komi_test.go:231-256, game/wq/game_test.go:9-20
    g = O.Game(O.KOMI, 3, 3, 3)
    assert g.eq(g)
    g3 = g.clone()
    g.apply(O.BLACK, 2)
    g.apply(O.WHITE, 4)
    g2 = g.clone()
    assert g.eq(g2)
    g.reset()
    assert g.eq(g3)
    w = O.Game(O.WQ, 19, 19, komi=7.5)
    w2 = w.clone()
    assert w.eq(w2)
    w.set_to_move(O.WHITE)
    assert not w.eq(w2)


@pytest.mark.parametrize("budget", KATS["mcts_example"]["budgets_reproducing_documented_game"])
def test_mcts_example_documented_game(budget):
    """mcts/example_test.go:74-156: `// Output: WINNER None` and the documented move list, with Timeout replaced
    by exactly `budget` iterations."""
    ex = O.ExampleSearch(O.MNK, 3, 3, 3, Budget=budget, inf=O.INF_SCRIPT)
    moves = []
    while True:
        best, ended, winner = ex.turn()
        moves.append(best)
        if ended:
            break
    assert moves == KATS["mcts_example"]["moves"]
    assert winner == KATS["mcts_example"]["winner"]
    assert ex.nn_evals() == 8 * budget + 9  # one eval per non-null simulation + one per prepareRoot


# SURVEY.md App. A "Validation of this appendix": root children after each Search at N = 200, produced by an
# INDEPENDENT transcription of the reference (the survey's throw-away Python); (move, visits, blackScores).
SURVEY_N200 = [
    [(4, 9, 3.0)] + [(m, 1, 0.0) for m in (0, 1, 2, 3, 5, 6, 7, 8)],
    [(0, 88, 50.0), (6, 45, 19.0), (7, 29, 12.0), (5, 13, 5.5), (1, 10, 4.5), (2, 10, 4.5), (3, 10, 4.5), (8, 10, 4.5)],
    [(2, 287, 211.0)] + [(m, 1, 0.0) for m in (1, 3, 5, 6, 7, 8)],
    [(6, 154, 139.5), (3, 88, 65.5), (1, 63, 48.5), (5, 62, 47.5), (7, 62, 47.5), (8, 62, 47.5)],
    [(3, 353, 337.5)] + [(m, 1, 0.0) for m in (1, 5, 7, 8)],
    [(5, 351, 347.0), (1, 102, 94.0), (7, 51, 47.0), (8, 51, 47.0)],
    [(1, 550, 547.0), (7, 1, 0.0), (8, 1, 0.0)],
    [(7, 702, 700.0), (8, 48, 46.0)],
    [(8, 901, 900.0)],
]


def test_mcts_example_root_statistics_n200():
    ex = O.ExampleSearch(O.MNK, 3, 3, 3, Budget=200, inf=O.INF_SCRIPT)
    gold = json.load(open(os.path.join(HERE, "golden", "oracle_mcts_example.json")))
    for t in range(9):
        best, ended, winner = ex.turn()
        kids = ex.root_children()
        assert kids == SURVEY_N200[t], "turn %d" % t
        assert kids == [tuple(c) for c in gold["turns"][t]["children"]]
        assert best == gold["turns"][t]["best"]
    assert ended and winner == 0


def test_dummy_inferer_closed_form():  # dummy.go:10-23 via the arena: uniform prior => first search visits spread evenly
    a = O.Arena(O.MNK, 3, 3, 3, Budget=9)
    a.set_inferencer(0, O.INF_DUMMY, dummy_player=1)
    a.set_inferencer(1, O.INF_DUMMY, dummy_player=1)
    a.begin(1)
    a.step()
    mv, vis, bs, pr = a.root_children(0)
    assert len(mv) == 9 and np.allclose(pr, 1.0 / 9.0)
    assert vis.sum() == 9 + 9 - 1 + 1 or vis.sum() >= 9  # every child starts at visits = 1 (tree.go:110)
