"""GPU parity, edge cases of the search/arena restated from the reference: zero and tiny budgets, resign thresholds and
DontResign, randomised move selection with temperature and a visit floor, odd board sizes, komi values that flip the
winner, the move cap.  Same bar as test_engine_gpu.py: bit-exact against the oracle."""
import numpy as np
import pytest

import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi
from test_engine_gpu import run_pair

pytestmark = pytest.mark.gpu


def test_budget_zero_moves_come_from_prepare_root_only(ctx):
    """Budget 0: Search runs prepareRoot (one evaluation, children with visits = 1) and bestMove (search.go:392-408,
    341-390) — priors decide."""
    dev, orcs = run_pair(ctx, capi.GAME_MNK, 3, 3, 3, budget=0, a_is_black=(1, 0))
    st = dev.stats()
    assert st["sims_total"] == 0 and st["nn_evals"] > 0


def test_budget_one_and_two(ctx):
    for b in (1, 2):
        run_pair(ctx, capi.GAME_C4, 6, 7, 4, budget=b, a_is_black=(1, 0), n_plies=12)


@pytest.mark.parametrize("pp", [capi.PREFER_PASS, capi.DONT_RESIGN])
def test_resignation_and_dont_resign(ctx, pp):
    """bestMove (search.go:341-390) turns a best-move Pass into Resign when shouldResign (search.go:502-530: past
    M*N/4 moves, score <= ResignPercentage) — unless PassPreference is DontResign.  5x5 wq, hash inferencer: the
    PreferPass game ends 9,8,19,...,11,Resign (White wins), the DontResign one with two passes (Black wins)."""
    dev, orcs = run_pair(ctx, capi.GAME_WQ, 5, 5, komi=0.5, enc=capi.ENC_WQ, budget=25, a_is_black=(1, 0), max_moves=60,
                         ResignPercentage=0.99, PassPreference=pp)
    for g in range(2):
        h = list(dev.history(g))
        if pp == capi.DONT_RESIGN:
            assert capi.RESIGN not in h and h[-2:] == [capi.PASS, capi.PASS]
            assert dev.game(g)[1]["winner"] == capi.BLACK
        else:
            assert h[-1] == capi.RESIGN
            assert dev.game(g)[1]["winner"] == capi.WHITE


def test_resign_percentage_is_inert_without_passes(ctx):
    """mnk has no Pass move, so no threshold can make it resign (the Resign replacement only applies to a Pass)."""
    dev, orcs = run_pair(ctx, capi.GAME_MNK, 3, 3, 3, budget=40, a_is_black=(1, 0), ResignPercentage=0.9)
    assert all(capi.RESIGN not in list(dev.history(g)) for g in range(2))


@pytest.mark.parametrize("temp,minv", [(0.5, 0), (2.0, 0), (1.0, 3)])
def test_randomised_selection_temperature_and_visit_floor(ctx, temp, minv):
    """randomizeChildren (tree.go:212-247): (visits/norm)^(1/T) weights, children under RandomMinVisits excluded."""
    S, G, budget = 99, 4, 50
    dev = A.Arena(ctx, capi.GAME_C4, 6, 7, 4, n_games=G, Budget=budget, seed=S, RandomCount=6, RandomTemperature=temp,
                  RandomMinVisits=minv)
    dev.set_inferencer(0, capi.INF_HASH)
    dev.set_inferencer(1, capi.INF_HASH)
    ab = np.array([1, 0, 1, 0], dtype=np.uint8)
    dev.reset(ab)
    dev.play(0, record=True)
    for g in range(G):
        o = O.Arena(O.C4, 6, 7, 4, Budget=budget, seed=S + g, RandomCount=6, RandomTemperature=temp, RandomMinVisits=minv)
        o.set_inferencer(0, O.INF_HASH)
        o.set_inferencer(1, O.INF_HASH)
        o.begin(int(ab[g]))
        o.play(0, True)
        np.testing.assert_array_equal(dev.history(g), o.history())
        assert dev.game(g)[1]["winner"] == o.state()[1]["winner"]


@pytest.mark.parametrize("size,komi", [(3, 0.5), (7, 0.5), (7, 30.5), (6, 5.5)])
def test_wq_odd_even_sizes_and_komi_values(ctx, size, komi):
    """wq boards of odd and even size; a large komi flips the winner of an otherwise Black-favoured game."""
    run_pair(ctx, capi.GAME_WQ, size, size, komi=komi, enc=capi.ENC_WQ, budget=16, a_is_black=(1, 0), max_moves=4 * size * size,
             n_plies=60)


def test_komi_game_sizes(ctx):
    for size in (3, 4, 6):
        run_pair(ctx, capi.GAME_KOMI, size, size, 3, budget=20, a_is_black=(1,), n_plies=50)


def test_move_cap_ends_the_game(ctx):
    """max_moves reached: the arena stops the game (declared cap; the reference loops until Ended())."""
    dev, orcs = run_pair(ctx, capi.GAME_WQ, 5, 5, komi=0.5, enc=capi.ENC_WQ, budget=8, a_is_black=(1, 0), max_moves=7)
    for g in range(2):
        assert len(dev.history(g)) <= 7


def test_mnk_rectangular_and_long_k(ctx):
    run_pair(ctx, capi.GAME_MNK, 3, 5, 3, budget=30, a_is_black=(1, 0))
    run_pair(ctx, capi.GAME_MNK, 4, 4, 4, budget=30, a_is_black=(1,))   # draws by full board


@pytest.mark.parametrize("dumb,pp", [(False, capi.DONT_PREFER_PASS), (False, capi.DONT_RESIGN), (True, capi.PREFER_PASS)])
def test_pass_policy_combinations(ctx, dumb, pp):
    """the three arms of bestMove's switch (search.go:369-385) on 5x5 wq, both colour assignments."""
    run_pair(ctx, capi.GAME_WQ, 5, 5, komi=0.5, enc=capi.ENC_WQ, budget=20, a_is_black=(1, 0), max_moves=50, DumbPass=dumb,
             PassPreference=pp)


def test_wq_randomised_opening_with_pass_child(ctx):
    S, G, budget = 7, 3, 20
    dev = A.Arena(ctx, capi.GAME_WQ, 5, 5, 0, 0.5, encoder=capi.ENC_WQ, n_games=G, Budget=budget, seed=S, RandomCount=8,
                  RandomTemperature=1.0, max_moves=40)
    dev.set_inferencer(0, capi.INF_HASH)
    dev.set_inferencer(1, capi.INF_HASH)
    ab = np.array([1, 0, 1], dtype=np.uint8)
    dev.reset(ab)
    dev.play(0, record=True)
    for g in range(G):
        o = O.Arena(O.WQ, 5, 5, 0, 0.5, enc=O.ENC_WQ, Budget=budget, seed=S + g, RandomCount=8, RandomTemperature=1.0, max_moves=40)
        o.set_inferencer(0, O.INF_HASH)
        o.set_inferencer(1, O.INF_HASH)
        o.begin(int(ab[g]))
        o.play(0, True)
        np.testing.assert_array_equal(dev.history(g), o.history())
        ob, op, ov = o.examples()
        dp, dpol, dval, dgi = dev.examples()
        sel = dgi == g
        np.testing.assert_array_equal(dp[sel].view(np.uint32), ob.view(np.uint32))
        np.testing.assert_array_equal(dval[sel], ov)


def test_more_games_than_compute_units(ctx):
    """700 concurrent tic-tac-toe games (more workgroups than CUs, several scheduling rounds): every game with the same
    colour assignment must be the same game, and both equal the oracle's."""
    G, budget = 700, 30
    ab = (np.arange(G) % 3 == 0).astype(np.uint8)
    dev = A.Arena(ctx, capi.GAME_MNK, 3, 3, 3, n_games=G, Budget=budget)
    dev.set_inferencer(0, capi.INF_HASH)
    dev.set_inferencer(1, capi.INF_HASH)
    dev.reset(ab)
    dev.play(0, record=True)
    want = {}
    for v in (0, 1):
        o = O.Arena(O.MNK, 3, 3, 3, Budget=budget)
        o.set_inferencer(0, O.INF_HASH)
        o.set_inferencer(1, O.INF_HASH)
        o.begin(v)
        o.play(0, True)
        want[v] = (list(o.history()), o.state()[1]["winner"], len(o.examples()[2]))
    _, _, _, gidx = dev.examples()
    counts = np.bincount(gidx, minlength=G)
    for g in range(G):
        h, w, ne = want[int(ab[g])]
        assert list(dev.history(g)) == h, g
        assert counts[g] == ne, g
    st = dev.stats()
    assert st["games_finished"] == G
    r = dev.results()
    assert r["a_wins"] + r["b_wins"] + r["draws"] == G


def test_full_pool_policy_strict_fails_stop_search_plays_on(ctx):
    """agz_arena_set_pool_policy.  A pool of ~4 expansions on 5x5 Go at Budget 40 fills in every search.  AGZ_POOL_STRICT (default): agz_arena_play
    fails with AGZ_E_TREE_FULL.  AGZ_POOL_STOP_SEARCH — the reference's MAXTREESIZE rule with max_nodes in its place (search.go:23,78,229: a full
    tree stops being searched for that move, the game goes on): every game runs to its end, tree_full counts the truncated searches, and every
    move played is legal in the oracle's rules; the single-tree handle returns the best move of the truncated search instead of failing."""
    G = 6
    kw = dict(encoder=capi.ENC_WQ, n_games=G, seed=5, Budget=40, max_nodes=110, max_moves=60)
    dev = A.Arena(ctx, capi.GAME_WQ, 5, 5, 0, 0.5, **kw)
    dev.set_inferencer(0, capi.INF_HASH)
    dev.set_inferencer(1, capi.INF_HASH)
    dev.reset()
    with pytest.raises(A.AgzError, match="overflowed"):
        dev.play(0, True)
    dev.close()
    dev = A.Arena(ctx, capi.GAME_WQ, 5, 5, 0, 0.5, **kw)
    dev.set_inferencer(0, capi.INF_HASH)
    dev.set_inferencer(1, capi.INF_HASH)
    dev.set_pool_policy(capi.POOL_STOP_SEARCH)
    dev.reset()
    dev.play(0, True)
    st = dev.stats()
    assert st["games_finished"] == G and st["n_active"] == 0 and st["tree_full"] > 0
    for g in range(G):
        og = O.Game(O.WQ, 5, 5, 0, 0.5)
        player = O.BLACK
        og.set_to_move(player)
        for mv in dev.history(g):
            if mv == capi.RESIGN:
                break
            assert mv == capi.PASS or og.check(player, int(mv)), (g, mv)
            og.apply(player, int(mv))
            player = O.WHITE if player == O.BLACK else O.BLACK
            og.set_to_move(player)
    with pytest.raises(A.AgzError):
        dev.set_pool_policy(7)
    dev.close()
    m = A.Mcts(ctx, capi.GAME_WQ, 5, 5, 0, 0.5, encoder=capi.ENC_WQ, Budget=40, max_nodes=110)
    m.set_inferencer(capi.INF_HASH)
    m.set_pool_policy(capi.POOL_STOP_SEARCH)
    og = O.Game(O.WQ, 5, 5, 0, 0.5)
    og.set_to_move(O.BLACK)
    m.set_game(board=og.board(), to_move=O.BLACK, n_moves=0, passes=0, hash=og.hash(), last_moves=[], historical=np.zeros((0, 25), np.int32))
    mv = m.search(O.BLACK)                  # no AgzError
    assert mv == capi.PASS or og.check(O.BLACK, mv)
    assert m.stats()["tree_full"] > 0
    m.close()
