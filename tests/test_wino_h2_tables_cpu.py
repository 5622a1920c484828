"""CPU: the Cook-Toom transform tables COMPILED into the WINO_H2 kernels (parsed from agogo_amd/csrc/conv_wino_h2.hpp, not
restated here) satisfy the Winograd identity exactly in rational arithmetic, in 1-D and in 2-D on ragged boards, and the range
bound the fp16x2 scaling relies on (|Bt d B| <= 2^VSHIFT max|d|) holds for them.  F(4x4,3x3): points 0, +-1, +-2, inf;
F(5x5,3x3): points 0, +-1, +-1/2, 2, inf."""
import os
import re
from fractions import Fraction

import numpy as np
import pytest

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "agogo_amd", "csrc", "conv_wino_h2.hpp")


def _num(tok):
    tok = tok.strip().rstrip("f")
    if "/" in tok:
        a, b = tok.split("/")
        return Fraction(_num(a)) / Fraction(_num(b))
    return Fraction(tok)


def _matrix(body, name):
    m = re.search(name + r"\[\d+\]\[\d+\]\s*=\s*\{(.*?)\};", body, re.S)
    assert m, name
    rows = re.findall(r"\{([^{}]*)\}", m.group(1))
    return [[_num(t) for t in r.split(",")] for r in rows]


def tables(tm):
    src = open(HDR).read()
    m = re.search(r"template <> struct WinoT<%d> \{(.*?)\n\};" % tm, src, re.S)
    assert m, "WinoT<%d> not found" % tm
    body = m.group(1)
    al = int(re.search(r"AL = (\d+)", body).group(1))
    vshift = int(re.search(r"VSHIFT = (\d+)", body).group(1))
    return al, vshift, _matrix(body, "BT"), _matrix(body, "AT"), _matrix(body, "G")


def matvec(M, v):
    return [sum(a * b for a, b in zip(row, v)) for row in M]


@pytest.mark.parametrize("tm", [4, 5])
def test_compiled_tables_satisfy_the_1d_identity_exactly(tm):
    al, vshift, BT, AT, G = tables(tm)
    assert al == tm + 2 and len(BT) == al and len(BT[0]) == al and len(AT) == tm and len(AT[0]) == al and len(G) == al and len(G[0]) == 3
    rng = np.random.default_rng(tm)
    for _ in range(20):
        d = [Fraction(int(v)) for v in rng.integers(-9, 10, al)]
        g = [Fraction(int(v)) for v in rng.integers(-9, 10, 3)]
        y = matvec(AT, [a * b for a, b in zip(matvec(G, g), matvec(BT, d))])
        ref = [d[i] * g[0] + d[i + 1] * g[1] + d[i + 2] * g[2] for i in range(tm)]
        assert y == ref          # exact: Fractions


@pytest.mark.parametrize("tm", [4, 5])
def test_range_bound_of_the_fp16x2_scaling(tm):
    """|V| = |Bt d B| <= (max row sum of |Bt|)^2 max|d| <= 2^VSHIFT max|d|: the per-board scale 2^(141 - VSHIFT - E(amax)) keeps |V s| < 2^15"""
    al, vshift, BT, AT, G = tables(tm)
    row = max(sum(abs(v) for v in r) for r in BT)
    assert row * row <= 2 ** vshift, (row, vshift)
    assert row * row > 2 ** (vshift - 1)       # and the shift is not wasteful


@pytest.mark.parametrize("tm,H,W", [(5, 19, 19), (5, 9, 9), (4, 6, 7), (5, 5, 5), (4, 19, 19), (5, 3, 4)])
def test_2d_tiling_with_ragged_edges_equals_direct_correlation(tm, H, W):
    """the kernels' tiling: tiles of tm x tm outputs over the zero-padded board, the last tile row / column hanging over the edge"""
    al, vshift, BT, AT, G = tables(tm)
    BTf, ATf, Gf = (np.array([[float(v) for v in r] for r in M]) for M in (BT, AT, G))
    rng = np.random.default_rng(10 * H + W)
    x = rng.integers(-4, 5, (H, W)).astype(np.float64)
    g = rng.integers(-4, 5, (3, 3)).astype(np.float64)
    nty, ntx = -(-H // tm), -(-W // tm)
    xp = np.zeros((nty * tm + 2, ntx * tm + 2))
    xp[1:H + 1, 1:W + 1] = x
    U = Gf @ g @ Gf.T
    y = np.zeros((nty * tm, ntx * tm))
    for ty in range(nty):
        for tx in range(ntx):
            d = xp[ty * tm:ty * tm + al, tx * tm:tx * tm + al]
            y[ty * tm:(ty + 1) * tm, tx * tm:(tx + 1) * tm] = ATf @ (U * (BTf @ d @ BTf.T)) @ ATf.T
    ref = np.zeros((H, W))
    for i in range(H):
        for j in range(W):
            ref[i, j] = (xp[i:i + 3, j:j + 3] * g).sum()
    np.testing.assert_allclose(y[:H, :W], ref, atol=1e-9)


def test_tile_choice_rule():
    """wino_h2_pick_tm: the tile with the fewer transform-domain rows (ties -> 4)"""
    def pick(H, W):
        r4 = 36 * (-(-H // 4)) * (-(-W // 4))
        r5 = 49 * (-(-H // 5)) * (-(-W // 5))
        return 5 if r5 < r4 else 4
    assert pick(19, 19) == 5 and pick(9, 9) == 5 and pick(6, 7) == 4 and pick(5, 5) == 5 and pick(13, 13) == 5 and pick(8, 8) == 4
    src = open(HDR).read()
    assert "return r5 < r4 ? 5 : 4;" in src
