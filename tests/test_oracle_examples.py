"""CPU: the oracle's example plumbing (oracle/examples.hpp) against the reference's own RotateBoard test and against
plain numpy restatements of shuffleExamples / prepareExamples."""
import numpy as np

import oracle_lib as O


def splitmix(seed):
    s = seed & (2**64 - 1)
    while True:
        s = (s + 0x9E3779B97F4A7C15) & (2**64 - 1)
        z = s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & (2**64 - 1)
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & (2**64 - 1)
        yield z ^ (z >> 31)


def test_rotate_board_reference_kat():
    """encoding_helper_test.go:10-55: the asymmetric 5x5 board returns to itself after 4 rotations (White=2, Black=1)."""
    W, B, N = 2.0, 1.0, 0.0
    board = np.array([W, N, N, N, B,
                      N, W, N, B, N,
                      N, N, N, N, N,
                      N, N, N, N, N,
                      B, N, N, N, W], np.float32)
    r = board
    seen = [board]
    for _ in range(4):
        r = O.rotate_board(r, 5, 5)
        seen.append(r)
    np.testing.assert_array_equal(seen[4], board)
    # the second row breaks the symmetry: the intermediate rotations all differ from the start
    for k in (1, 2, 3):
        assert not np.array_equal(seen[k], board)
    # one application is new[i][j] = old[j][m-1-i] (the 4-cycle of encoding_helper.go:92-103) = numpy's rot90 (ccw)
    np.testing.assert_array_equal(seen[1].reshape(5, 5), np.rot90(board.reshape(5, 5), 1))
    assert O.rotate_board(np.zeros(6, np.float32), 2, 3) is None   # "only takes square boards"


def test_rotate_board_even_and_odd_sizes():
    rng = np.random.default_rng(0)
    for m in (1, 2, 3, 4, 9, 19):
        b = rng.normal(size=(m, m)).astype(np.float32)
        np.testing.assert_array_equal(O.rotate_board(b, m, m).reshape(m, m), np.rot90(b, 1))


def test_augment_and_prepare_match_numpy_restatement():
    F, m, A1, n = 3, 4, 17, 23
    rng = np.random.default_rng(1)
    boards = rng.normal(size=(n, F, m, m)).astype(np.float32)
    pol = rng.random((n, A1)).astype(np.float32)
    val = rng.choice(np.array([-1, 0, 1], np.float32), n)
    s = O.ExampleSet(F, m, m, A1)
    s.push(boards, pol, val)
    assert s.augment_rotate()
    b4, p4, v4 = s.get()
    assert len(s) == 4 * n
    for e in (0, 7, 22):
        for q in range(4):
            np.testing.assert_array_equal(b4[4 * e + q].reshape(F, m, m), np.rot90(boards[e], q, axes=(1, 2)))
            np.testing.assert_array_equal(p4[4 * e + q][:m * m].reshape(m, m), np.rot90(pol[e][:m * m].reshape(m, m), q))
            assert p4[4 * e + q][m * m] == pol[e][m * m] and v4[4 * e + q] == val[e]
    # prepareExamples with a maxExamples cut: two Fisher-Yates passes from ONE SplitMix64 stream
    BS, mx, seed = 8, 50, 99
    idx = list(range(4 * n))
    g = splitmix(seed)
    for i in range(len(idx)):
        j = next(g) % (i + 1)
        idx[i], idx[j] = idx[j], idx[i]
    idx = idx[:mx]
    for i in range(len(idx)):
        j = next(g) % (i + 1)
        idx[i], idx[j] = idx[j], idx[i]
    batches, X, P, V = s.prepare(BS, mx, seed)
    assert batches == mx // BS and X.shape[0] == batches * BS
    want = idx[:batches * BS]
    np.testing.assert_array_equal(X.reshape(len(want), -1), b4[want])
    np.testing.assert_array_equal(P, p4[want])
    np.testing.assert_array_equal(V, v4[want])
