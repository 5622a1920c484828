"""GPU: device example plumbing (agz_examples_*, agz_rotate_boards, agz_train_dev) vs the oracle — bit-exact (pure data
movement + integer index arithmetic); agz_train_dev vs agz_train within the trainer's fp32 tolerance."""
import numpy as np
import pytest

import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m", [1, 2, 5, 9, 19])
def test_rotate_boards_match_oracle(ctx, m):
    rng = np.random.default_rng(m)
    b = rng.normal(size=(7, m, m)).astype(np.float32)
    got = A.rotate_boards(ctx, b, m, m)
    for i in range(7):
        np.testing.assert_array_equal(got[i].ravel(), O.rotate_board(b[i], m, m).ravel())


def test_rotate_boards_reference_kat_and_error(ctx):
    W, B, N = 2.0, 1.0, 0.0   # encoding_helper_test.go:10-55
    board = np.array([W, N, N, N, B, N, W, N, B, N, N, N, N, N, N, N, N, N, N, N, B, N, N, N, W], np.float32)
    r = board
    for _ in range(4):
        r = A.rotate_boards(ctx, r, 5, 5)
    np.testing.assert_array_equal(r, board)
    with pytest.raises(A.AgzError, match="only takes square boards"):
        A.rotate_boards(ctx, np.zeros(6, np.float32), 2, 3)


@pytest.mark.parametrize("F,m,A1,n,BS,mx", [(2, 3, 10, 37, 8, 0), (18, 9, 82, 50, 16, 120), (3, 4, 17, 23, 8, 50)])
def test_augment_prepare_bit_exact_vs_oracle(ctx, F, m, A1, n, BS, mx):
    rng = np.random.default_rng(n)
    boards = rng.normal(size=(n, F, m, m)).astype(np.float32)
    pol = rng.random((n, A1)).astype(np.float32)
    val = rng.choice(np.array([-1, 0, 1], np.float32), n)
    ex = A.Examples(ctx, F, m, m, A1)
    ex.append_host(boards[:10], pol[:10], val[:10])      # two appends: exercises the growth path
    ex.append_host(boards[10:], pol[10:], val[10:])
    o = O.ExampleSet(F, m, m, A1)
    o.push(boards, pol, val)
    assert len(ex) == n
    ex.augment_rotate()
    assert o.augment_rotate()
    assert len(ex) == 4 * n == len(o)
    for got, want in zip(ex.get(), o.get()):
        np.testing.assert_array_equal(got, want)
    batches = ex.prepare(BS, mx, seed=4242)
    ob, oX, oP, oV = o.prepare(BS, mx, seed=4242)
    assert batches == ob
    X, P, V = ex.tensors()
    np.testing.assert_array_equal(X, oX)
    np.testing.assert_array_equal(P, oP)
    np.testing.assert_array_equal(V, oV)
    assert ex.tensors_dev()[3] == batches * BS


def test_non_square_augment_is_rejected_and_too_few_examples_give_zero_batches(ctx):
    ex = A.Examples(ctx, 2, 6, 7, 8)
    ex.append_host(np.zeros((3, 2, 6, 7), np.float32), np.zeros((3, 8), np.float32), np.zeros(3, np.float32))
    with pytest.raises(A.AgzError, match="only takes square boards"):
        ex.augment_rotate()
    assert ex.prepare(256) == 0       # agogo.go:123-125 "batches is nil"
    ex.clear()
    assert len(ex) == 0 and ex.prepare(1) == 0


def test_append_arena_is_reference_order(ctx):
    """agogo.go:110-114: episodes appended one after another, each in ply order (device records in completion order)."""
    G = 12
    arena = A.Arena(ctx, capi.GAME_MNK, 3, 3, 3, encoder=capi.ENC_TWOPLANE, n_games=G, seed=5, Budget=30)
    arena.set_inferencer(0, capi.INF_HASH)
    arena.set_inferencer(1, capi.INF_HASH)
    arena.reset()
    arena.play(0, True)
    planes, policy, value, gidx = arena.examples()
    assert len(value) > G
    order = np.argsort(gidx, kind="stable")
    ex = A.Examples(ctx, 2, 3, 3, 10)
    ex.append_arena(arena)
    p, q, v = ex.get()
    np.testing.assert_array_equal(p, planes[order])
    np.testing.assert_array_equal(q, policy[order])
    np.testing.assert_array_equal(v, value[order])
    assert set(np.unique(v)) <= {-1.0, 0.0, 1.0}     # labelled (arena.go:146-155)


def test_train_dev_matches_host_train(ctx):
    """agz_train_dev (row-index shuffle, device gathers) == agz_train (in-place row swaps on the host) for the same seed."""
    K, L, FC, S, F, Aspace, B = 32, 1, 16, 3, 2, 10, 8
    rng = np.random.default_rng(3)
    rows = 3 * B
    X = rng.choice(np.array([0.0, 1.0], np.float32), size=(rows, F, S, S)).astype(np.float32)
    P = np.zeros((rows, Aspace), np.float32)
    P[np.arange(rows), rng.integers(0, Aspace, rows)] = 1
    V = rng.choice(np.array([-1, 0, 1], np.float32), rows)
    t1 = A.Trainer(ctx, K, L, FC, S, S, F, Aspace, B)
    t2 = A.Trainer(ctx, K, L, FC, S, S, F, Aspace, B)
    t1.init_random(9)
    t2.init_random(9)
    ex2 = A.Examples(ctx, F, S, S, Aspace)
    ex2.append_host(X, P, V)
    b = ex2.prepare(B, 0, seed=1)
    Xp, Pp, Vp = ex2.tensors()
    ch = t1.train(Xp.copy(), Pp.copy(), Vp.copy(), b, 4, seed=77)
    xd, pd, vd, rows_d, bd = ex2.tensors_dev()
    assert rows_d == rows and bd == 3
    cd = t2.train_dev(xd, pd, vd, bd, 4, seed=77)
    assert abs(cd - ch) <= 1e-4 * max(1.0, abs(ch))
    for i in range(t2.num_params()):
        a, bb = t1.get_param(i), t2.get_param(i)
        assert np.abs(a - bb).max() <= 1e-4 * max(np.abs(a).max(), 1e-3), t2.param_info(i)
    # device tensors are left in place
    X2, P2, V2 = ex2.tensors()
    np.testing.assert_array_equal(X2, Xp)


def test_append_arena_takes_the_finished_games_and_keeps_the_running_ones(ctx):
    """continuous self-play harvested repeatedly.  agz_examples_append_arena TAKES the rows of finished games
    out of the arena (a second call appends nothing new), the rows of games still in flight stay — with their earlier plies — and
    come out labelled once their game ends.  Checked against a twin arena (same seed, same calls) that is never harvested: after
    every harvest, rows appended so far + rows left in the arena are exactly the twin's rows — same planes, policies AND labels,
    i.e. the per-game chains survived the compaction and every game was labelled as a whole."""
    G = 16

    def make():
        a = A.Arena(ctx, capi.GAME_MNK, 3, 3, 3, encoder=capi.ENC_TWOPLANE, n_games=G, seed=9, Budget=20)
        a.set_inferencer(0, capi.INF_HASH)
        a.set_inferencer(1, capi.INF_HASH)
        a.reset()
        return a

    def rows(p, q, v):
        m = np.concatenate([np.asarray(p).reshape(len(v), 18), np.asarray(q).reshape(len(v), 10), np.asarray(v).reshape(-1, 1)], axis=1)
        return m[np.lexsort(m.T[::-1])] if len(v) else m

    arena, twin = make(), make()
    ex = A.Examples(ctx, 2, 3, 3, 10)
    for a in (arena, twin):
        a.play(4, True)                       # four plies of every game: nothing has ended, nothing to take
    ex.append_arena(arena)
    assert len(ex) == 0 and len(arena.examples()[2]) == 4 * G
    for more in (5, 11, 30):
        for a in (arena, twin):
            a.selfplay(a.stats()["games_finished"] + more, record=True)   # finished games restart at once: some are always mid-way
        before = len(ex)
        ex.append_arena(arena)
        assert len(ex) > before
        again = len(ex)
        ex.append_arena(arena)                # nothing finished in between: no duplicates
        assert len(ex) == again
        left = arena.examples()
        st = arena.stats()
        assert st["examples_dropped"] == 0 and st == twin.stats()
        assert len(ex) + len(left[2]) == st["examples"]
        assert set(np.unique(left[2])) <= {1.0, 2.0}          # what stays is unlabelled (the raw mover colour)
        p, q, v = ex.get()
        assert set(np.unique(v)) <= {-1.0, 0.0, 1.0}
        tp, tq, tv, _ = twin.examples()
        mine = rows(np.concatenate([p.reshape(len(v), 18), left[0].reshape(len(left[2]), 18)]),
                    np.concatenate([q, left[1]]), np.concatenate([v, left[2]]))
        np.testing.assert_array_equal(mine, rows(tp, tq, tv))
