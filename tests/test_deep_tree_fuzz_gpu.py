"""GPU parity, randomised configurations of the NARROW-TREE regime (fixed seeds): game kind, board size, encoder, budget, lanes, how peaked the
policy is (0.5 .. 0.99 on one action), whether the favourite is drawn among the empty points (legal most of the time: long principal lines)
or over all actions, how large the value signal is (0: priors alone steer; 0.5: Q decides), pass / resign policy, randomised openings —
device (AGZ_INF_CALLBACK) vs oracle (its own callback) on the SAME host inferencer, bit-exact: roots after every ply, histories, examples.
The hand-written cases of test_deep_tree_gpu.py pin the depth cap and the in-tree terminals; this sweeps the interactions around them."""
import os
import zlib

import numpy as np
import pytest

from conftest import fuzz_seeds

import oracle_lib as O
from agogo_amd import capi
from test_deep_tree_gpu import f32bits, run

pytestmark = pytest.mark.gpu
KINDS = {capi.GAME_MNK: O.MNK, capi.GAME_C4: O.C4, capi.GAME_KOMI: O.KOMI, capi.GAME_WQ: O.WQ}


def peaked(plen, cells, peak, value_amp, among_empty, twoplane):
    """`peak` on one action chosen by a CRC of the planes (among the EMPTY points of plane 0 when the two-plane encoder shows the board and
    among_empty is set), the rest spread evenly; value: a CRC-derived number in [-value_amp, value_amp]"""
    def one(planes_flat):
        x = np.ascontiguousarray(planes_flat, np.float32)
        h = zlib.crc32(x.tobytes())
        p = np.full(plen, (1.0 - peak) / (plen - 1), np.float32)
        fav = h % plen
        if among_empty and twoplane:
            empt = np.where(np.abs(x[:cells]) < 0.5)[0]
            if len(empt) and plen > cells:          # (c4: one action per column — the hash picks the column as it stands)
                fav = int(empt[h % len(empt)])
        p[fav] = peak
        return p, float(np.float32(value_amp * (((h >> 8) % 2001) - 1000) / 1000.0))
    return one


WIDE = bool(os.environ.get("AGZ_FUZZ_WIDE"))   # soak runs: bigger boards, budgets, more lanes and games


def draw(rng):
    c = _draw(rng)
    if WIDE:
        if c["kind"] == capi.GAME_WQ:
            s = int(rng.choice([5, 7, 9, 13, 19]))
            c.update(m=s, n=s)
        c["budget"] = int(rng.choice([30, 120, 300, 600]))
        c["lanes"] = int(rng.choice([1, 1, 2, 3, 7, 16]))
        c["G"] = int(rng.integers(1, 9))
        c["peak"] = float(rng.choice([0.5, 0.8, 0.95, 0.99, 0.999]))
    return c


def _draw(rng):
    kind = int(rng.choice([capi.GAME_MNK, capi.GAME_C4, capi.GAME_KOMI, capi.GAME_WQ, capi.GAME_WQ]))
    c = dict(kind=kind, k=0, komi=0.0)
    if kind == capi.GAME_MNK:
        m, n = int(rng.integers(3, 6)), int(rng.integers(3, 6))
        c.update(m=m, n=n, k=int(rng.integers(3, min(m, n) + 1)), enc=capi.ENC_TWOPLANE)
    elif kind == capi.GAME_C4:
        c.update(m=6, n=7, k=4, enc=capi.ENC_TWOPLANE)
    elif kind == capi.GAME_KOMI:
        s = int(rng.integers(4, 7))
        c.update(m=s, n=s, k=int(rng.integers(2, 5)), enc=capi.ENC_TWOPLANE)
    else:
        s = int(rng.choice([4, 5, 6, 7, 9]))
        c.update(m=s, n=s, komi=float(rng.choice([0.5, 5.5, 7.5])), enc=int(rng.choice([capi.ENC_WQ, capi.ENC_TWOPLANE])))
    c["budget"] = int(rng.choice([8, 30, 80, 160, 300]))
    c["lanes"] = int(rng.choice([1, 1, 1, 2, 4, 8]))
    c["peak"] = float(rng.choice([0.5, 0.8, 0.95, 0.99]))
    c["value_amp"] = float(rng.choice([0.0, 0.0, 0.05, 0.5]))
    c["among_empty"] = bool(rng.integers(0, 2))
    c["G"] = int(rng.integers(1, 5))
    c["plies"] = int(rng.integers(2, 10))
    c["DumbPass"] = bool(rng.integers(0, 2))
    c["PassPreference"] = int(rng.choice([capi.DONT_PREFER_PASS, capi.PREFER_PASS, capi.DONT_RESIGN]))
    c["ResignPercentage"] = float(rng.choice([0.0, 0.0, 0.3]))
    c["PUCT"] = float(rng.choice([1.0, 0.5, 0.25]))
    c["RandomCount"] = int(rng.choice([0, 0, 4]))
    return c


@pytest.mark.parametrize("seed", fuzz_seeds(160))
def test_random_narrow_tree_configuration(ctx, seed):
    rng = np.random.default_rng(7000 + seed)
    c = draw(rng)
    cells = c["m"] * c["n"]
    A_ = c["n"] if c["kind"] == capi.GAME_C4 else cells
    plen = A_ + 1
    one = peaked(plen, cells, c["peak"], c["value_amp"], c["among_empty"], c["enc"] == capi.ENC_TWOPLANE)
    openings = [int(x) for x in rng.integers(0, max(1, cells // 3), size=c["G"])]
    grow = bool(seed & 1)
    dev, orcs = run(ctx, c["kind"], KINDS[c["kind"]], c["m"], c["n"], c["komi"], c["enc"], one, plen, G=c["G"], budget=c["budget"], plies=c["plies"],
                    openings=openings, lanes=c["lanes"], PassPreference=c["PassPreference"], max_moves=3 * cells, k=c["k"], DumbPass=c["DumbPass"],
                    # (pools for EVERY search's expansions: a 0.999-peaked tree keeps nearly all its nodes move after move — the default, four
                    #  searches' worth, is a sizing rule for kept fractions up to 3/4, include/agz.h; parity is what is tested here)
                    # ... or, every other configuration, pools of ONE search's worth that grow (AGZ_POOL_GROW): the same results
                    max_nodes=((c["plies"] + 2) if not grow else 1) * (c["budget"] + 2) * (plen + 1), pool_policy=capi.POOL_GROW if grow else None,
                    ResignPercentage=c["ResignPercentage"], PUCT=c["PUCT"], RandomCount=c["RandomCount"], RandomTemperature=1.0, RandomMinVisits=0)
    dp, dpol, dval, dgi = dev.examples()
    for g in range(c["G"]):
        np.testing.assert_array_equal(dev.history(g), orcs[g].history())
        ob, op, ov = orcs[g].examples()
        sel = dgi == g
        assert sel.sum() == ob.shape[0], (g, c)
        if ob.shape[0]:
            np.testing.assert_array_equal(f32bits(dp[sel]), f32bits(ob))
            np.testing.assert_array_equal(f32bits(dpol[sel]), f32bits(op))
    st = dev.stats()
    assert st["tree_full"] == 0
    tot_e = sum(o.tree_stats(a)["nn_evals"] for o in orcs for a in (0, 1))
    tot_p = sum(o.tree_stats(a)["playouts"] for o in orcs for a in (0, 1))
    assert st["sims_nonnull"] == tot_p, c
    if c["lanes"] == 1:
        assert st["nn_evals"] == tot_e, c
    dev.close()
