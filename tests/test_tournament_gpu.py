"""GPU parity: tournament-style use — one searching agent against an EXTERNAL opponent whose moves are injected with
agz_arena_apply_moves (the device-side equivalent of handing Agent.Search a game.State the caller advanced itself,
agent.go:76-81).  The agent's tree must re-root over the two plies played since its last search (updateRoot,
search.go:424-500) exactly like the oracle's."""
import numpy as np
import pytest

from conftest import fuzz_seeds

import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi

pytestmark = pytest.mark.gpu


def external_reply(board, kind, m, n, k_ply):
    """a deterministic scripted opponent: the (k_ply-th mod count) empty cell / open column"""
    b = np.asarray(board).reshape(m, n)
    if kind == capi.GAME_C4:
        cols = [c for c in range(n) if b[0, c] == 0]
        return cols[k_ply % len(cols)] if cols else capi.PASS
    empt = np.flatnonzero(b.ravel() == 0)
    return int(empt[(3 * k_ply + 1) % len(empt)]) if len(empt) else capi.PASS


@pytest.mark.parametrize("kind,m,n,k,enc,komi", [
    (capi.GAME_MNK, 3, 3, 3, capi.ENC_TWOPLANE, 0.0),
    (capi.GAME_MNK, 5, 5, 4, capi.ENC_TWOPLANE, 0.0),
    (capi.GAME_WQ, 5, 5, 0, capi.ENC_WQ, 0.5),
    (capi.GAME_KOMI, 5, 5, 3, capi.ENC_TWOPLANE, 0.0),
])
def test_agent_vs_external_opponent_matches_oracle(ctx, kind, m, n, k, enc, komi):
    G, budget = 2, 40
    ab = np.array([1, 0], dtype=np.uint8)           # game 0: the agent (A) is Black and moves first; game 1: the outsider does
    searched = 0
    # The arena advances ALL unfinished games per call, so mixed plies (one game searching, the other injected) need one
    # arena per phase class: run the two colour assignments as two single-game arenas.
    for g in range(G):
        dev1 = A.Arena(ctx, kind, m, n, k, komi, encoder=enc, n_games=1, Budget=budget, max_moves=60, seed=1337 + g)
        dev1.set_inferencer(0, capi.INF_HASH)
        dev1.set_inferencer(1, capi.INF_HASH)
        dev1.reset(ab[g:g + 1])
        o = O.Arena(kind, m, n, k, komi, enc=enc, Budget=budget, max_moves=60)
        o.set_inferencer(0, O.INF_HASH)
        o.set_inferencer(1, O.INF_HASH)
        o.begin(int(ab[g]))
        for ply in range(30):
            if o.state()[1]["ended"]:
                break
            a_to_move = (ply % 2 == 0) == bool(ab[g])
            if a_to_move:
                dev1.begin_move()
                dev1.simulate(budget)
                dev1.end_move(True)
                o.step(True)
                omv, ovis, obs, opr = o.root_children(0)
                dmv, dvis, dbs, dpr = dev1.root_children(0, 0)
                np.testing.assert_array_equal(dmv, omv, err_msg="game %d ply %d" % (g, ply))
                np.testing.assert_array_equal(dvis, ovis, err_msg="game %d ply %d" % (g, ply))
                np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32))
                searched += 1
            else:
                board, _ = dev1.game(0)
                # the outsider: first candidate (scripted order) the rules accept; the oracle's Check is the referee,
                # and a rejected candidate must be rejected by the device too
                empt = list(np.flatnonzero(np.asarray(board).ravel() == 0))
                start = (3 * ply + 1) % max(len(empt), 1)
                cands = [external_reply(board, kind, m, n, ply)] if kind == capi.GAME_C4 else empt[start:] + empt[:start]
                mv = None
                for cnd in cands:
                    r = o.apply_move(int(cnd))
                    if r >= 0:
                        mv = int(cnd)
                        break
                    with pytest.raises(A.AgzError, match="illegal"):
                        dev1.apply_moves(np.array([cnd], dtype=np.int32))
                if mv is None:
                    mv = capi.PASS
                    assert o.apply_move(mv) >= 0
                dev1.apply_moves(np.array([mv], dtype=np.int32))
            ob, ost = o.state()
            db, dst = dev1.game(0)
            np.testing.assert_array_equal(db, ob)
            assert dst["ended"] == ost["ended"] and (not ost["ended"] or dst["winner"] == ost["winner"])
            np.testing.assert_array_equal(dev1.history(0), o.history())
        # examples: only the agent's searched plies were recorded
        dp, dpol, dval, _ = dev1.examples()
        ob_, op_, ov_ = o.examples()
        np.testing.assert_array_equal(dp.view(np.uint32), ob_.view(np.uint32))
        np.testing.assert_array_equal(dpol.view(np.uint32), op_.view(np.uint32))
    assert searched >= 6


def test_illegal_external_move_is_rejected(ctx):
    dev = A.Arena(ctx, capi.GAME_MNK, 3, 3, 3, n_games=2, Budget=5)
    dev.set_inferencer(0, capi.INF_HASH)
    dev.set_inferencer(1, capi.INF_HASH)
    dev.reset(np.array([1, 1], dtype=np.uint8))
    dev.apply_moves(np.array([4, 0], dtype=np.int32))
    with pytest.raises(A.AgzError, match="illegal"):
        dev.apply_moves(np.array([4, 1], dtype=np.int32))     # game 0: cell 4 is taken; game 1 is applied
    assert list(dev.history(0)) == [4] and list(dev.history(1)) == [0, 1]
    with pytest.raises(A.AgzError, match="illegal"):
        dev.apply_moves(np.array([capi.PASS, 9], dtype=np.int32))   # mnk has no pass; 9 is off the board
    assert list(dev.history(0)) == [4] and list(dev.history(1)) == [0, 1]
    dev.apply_moves(np.array([0, capi.NO_MOVE], dtype=np.int32))      # only game 0 advances
    assert list(dev.history(0)) == [4, 0] and list(dev.history(1)) == [0, 1]
    dev.begin_move()
    with pytest.raises(A.AgzError, match="in progress"):
        dev.apply_moves(np.array([0, 2], dtype=np.int32))


@pytest.mark.parametrize("seed", fuzz_seeds(16))
def test_tournament_fuzz(ctx, seed):
    """random game / size / budget / lanes / pass policy; the outsider plays random candidate moves, the oracle's Check is
    the referee and the device must accept and reject exactly the same candidates; trees, boards, move lists bit-exact."""
    rng = np.random.default_rng(7000 + seed)
    kind = int(rng.choice([capi.GAME_MNK, capi.GAME_KOMI, capi.GAME_WQ]))
    s = int(rng.integers(3, 7))
    k = int(rng.integers(3, min(s, 4) + 1)) if kind != capi.GAME_WQ else 0
    enc = capi.ENC_WQ if kind == capi.GAME_WQ else capi.ENC_TWOPLANE
    komi = 0.5 if kind == capi.GAME_WQ else 0.0
    budget = int(rng.choice([1, 5, 20, 40]))
    lanes = int(rng.choice([1, 1, 3, 8]))
    a_black = int(rng.integers(0, 2))
    kw = dict(DumbPass=bool(rng.integers(0, 2)), PassPreference=int(rng.choice([capi.DONT_PREFER_PASS, capi.PREFER_PASS])))
    dev = A.Arena(ctx, kind, s, s, k, komi, encoder=enc, n_games=1, Budget=budget, max_moves=3 * s * s, **kw)
    dev.set_inferencer(0, capi.INF_HASH)
    dev.set_inferencer(1, capi.INF_HASH)
    dev.set_parallel(lanes)
    dev.reset(np.array([a_black], dtype=np.uint8))
    o = O.Arena(kind, s, s, k, komi, enc=enc, Budget=budget, max_moves=3 * s * s, **kw)
    o.set_inferencer(0, O.INF_HASH)
    o.set_inferencer(1, O.INF_HASH)
    o.set_parallel(lanes)
    o.begin(a_black)
    for ply in range(24):
        if o.state()[1]["ended"]:
            break
        a_to_move = (ply % 2 == 0) == bool(a_black)
        if a_to_move:
            dev.begin_move()
            dev.simulate(budget)
            dev.end_move(True)
            o.step(True)
            omv, ovis, obs, _ = o.root_children(0)
            dmv, dvis, dbs, _ = dev.root_children(0, 0)
            np.testing.assert_array_equal(dmv, omv, err_msg="ply %d" % ply)
            np.testing.assert_array_equal(dvis, ovis, err_msg="ply %d" % ply)
            np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32))
        else:
            cands = list(rng.permutation(s * s)[:6]) + [capi.PASS]
            played = False
            for cnd in cands:
                r = o.apply_move(int(cnd))
                if r >= 0:
                    dev.apply_moves(np.array([cnd], dtype=np.int32))
                    played = True
                    break
                with pytest.raises(A.AgzError, match="illegal"):
                    dev.apply_moves(np.array([cnd], dtype=np.int32))
            if not played:   # nothing legal among the candidates (mnk/komi have no pass): resign
                assert o.apply_move(capi.RESIGN) >= 0
                dev.apply_moves(np.array([capi.RESIGN], dtype=np.int32))
        ob, ost = o.state()
        db, dst = dev.game(0)
        np.testing.assert_array_equal(db, ob)
        assert dst["ended"] == ost["ended"] and (not ost["ended"] or dst["winner"] == ost["winner"])
        np.testing.assert_array_equal(dev.history(0), o.history())


@pytest.mark.parametrize("kind,s,seed", [(capi.GAME_WQ, 9, 0), (capi.GAME_WQ, 7, 1), (capi.GAME_WQ, 13, 2), (capi.GAME_KOMI, 7, 3),
                                         (capi.GAME_KOMI, 9, 4), (capi.GAME_WQ, 5, 5)])
def test_go_rules_long_random_rollout(ctx, kind, s, seed):
    """Both sides' moves injected at random (the oracle's Check referees; rejected candidates must be rejected by the device
    too) for up to 160 plies: dense boards, captures, merges, the reference's suicide rule — board, captures-so-far,
    end state compared after every ply, a search every 16 plies compared tree for tree."""
    rng = np.random.default_rng(8000 + seed)
    k = 0 if kind == capi.GAME_WQ else 200          # komi: capture target high enough not to end the game early
    enc = capi.ENC_WQ if kind == capi.GAME_WQ else capi.ENC_TWOPLANE
    komi = 6.5 if kind == capi.GAME_WQ else 0.0
    budget = 6
    dev = A.Arena(ctx, kind, s, s, k, komi, encoder=enc, n_games=1, Budget=budget, max_moves=400)
    dev.set_inferencer(0, capi.INF_HASH)
    dev.set_inferencer(1, capi.INF_HASH)
    dev.reset(np.array([1], dtype=np.uint8))
    o = O.Arena(kind, s, s, k, komi, enc=enc, Budget=budget, max_moves=400)
    o.set_inferencer(0, O.INF_HASH)
    o.set_inferencer(1, O.INF_HASH)
    o.begin(1)
    rejected = 0
    for ply in range(160):
        _, ost = o.state()
        if ost["ended"]:
            break
        if ply % 16 == 15:
            agent = 0 if ost["to_move"] == O.BLACK else 1
            dev.begin_move()
            dev.simulate(budget)
            dev.end_move(False)
            o.step(False)
            omv, ovis, obs, _ = o.root_children(agent)
            dmv, dvis, dbs, _ = dev.root_children(0, agent)
            np.testing.assert_array_equal(dmv, omv, err_msg="ply %d" % ply)
            np.testing.assert_array_equal(dvis, ovis, err_msg="ply %d" % ply)
            np.testing.assert_array_equal(dbs.view(np.uint32), obs.view(np.uint32))
        else:
            played = False
            for cnd in rng.permutation(s * s)[:12]:
                r = o.apply_move(int(cnd))
                if r >= 0:
                    dev.apply_moves(np.array([cnd], dtype=np.int32))
                    played = True
                    break
                rejected += 1
                with pytest.raises(A.AgzError, match="illegal"):
                    dev.apply_moves(np.array([cnd], dtype=np.int32))
            if not played:
                mv = capi.PASS if kind == capi.GAME_WQ else capi.RESIGN
                assert o.apply_move(mv) >= 0
                dev.apply_moves(np.array([mv], dtype=np.int32))
        ob, ost = o.state()
        db, dst = dev.game(0)
        np.testing.assert_array_equal(db, ob, err_msg="board after ply %d" % ply)
        assert dst["ended"] == ost["ended"] and (not ost["ended"] or dst["winner"] == ost["winner"]), ply
        np.testing.assert_array_equal(dev.history(0), o.history())
    assert rejected > 0      # occupied points / suicides were met: the rejection path was exercised
