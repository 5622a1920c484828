"""GPU parity, randomised configurations (fixed seeds): game kind, board size, budget, pass/resign policy, randomised
opening (count, temperature, visit floor), inferencer, lanes, colour assignment, whole games on small boards — device vs oracle, bit-exact, a handful of plies each.  Catches
interactions the hand-written cases do not enumerate."""
import os

import numpy as np
import pytest

from conftest import fuzz_seeds

import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi
from test_engine_gpu import run_pair

pytestmark = pytest.mark.gpu


WIDE = bool(os.environ.get("AGZ_FUZZ_WIDE"))   # soak runs: a broader distribution (bigger boards, budgets, lanes, more games)


def draw_config(rng):
    kind = int(rng.choice([capi.GAME_MNK, capi.GAME_C4, capi.GAME_KOMI, capi.GAME_WQ]))
    cfg = dict(kind=kind)
    if WIDE:
        return draw_wide(rng, cfg)
    if kind == capi.GAME_MNK:
        m, n = int(rng.integers(3, 6)), int(rng.integers(3, 6))
        cfg.update(m=m, n=n, k=int(rng.integers(3, min(m, n) + 1)), komi=0.0, enc=capi.ENC_TWOPLANE)
    elif kind == capi.GAME_C4:
        cfg.update(m=6, n=7, k=4, komi=0.0, enc=capi.ENC_TWOPLANE)
    elif kind == capi.GAME_KOMI:
        s = int(rng.integers(3, 7))
        cfg.update(m=s, n=s, k=int(rng.integers(1, 4)), komi=0.0, enc=capi.ENC_TWOPLANE)
    else:
        s = int(rng.integers(3, 8))
        cfg.update(m=s, n=s, k=0, komi=float(rng.choice([0.5, 5.5, 7.5])), enc=capi.ENC_WQ)
    cfg["budget"] = int(rng.choice([1, 3, 10, 25, 60]))
    cfg["inf"] = int(rng.choice([capi.INF_HASH, capi.INF_HASH, capi.INF_DUMMY, capi.INF_UNIFORM]))
    cfg["parallel"] = int(rng.choice([1, 1, 2, 5, 8]))
    cfg["a_is_black"] = tuple(int(x) for x in rng.integers(0, 2, size=int(rng.integers(1, 4))))
    cfg["DumbPass"] = bool(rng.integers(0, 2))
    cfg["PassPreference"] = int(rng.choice([capi.DONT_PREFER_PASS, capi.PREFER_PASS, capi.DONT_RESIGN]))
    cfg["ResignPercentage"] = float(rng.choice([0.0, 0.0, 0.3, -1.0]))
    cfg["PUCT"] = float(rng.choice([1.0, 0.5, 0.25]))
    cfg["RandomCount"] = int(rng.choice([0, 0, 3, 6]))
    cfg["RandomTemperature"] = float(rng.choice([1.0, 0.7, 1.5]))
    cfg["RandomMinVisits"] = int(rng.choice([0, 0, 2]))
    cfg["full_game"] = bool(rng.integers(0, 3) == 0) and cfg["m"] * cfg["n"] <= 25
    return cfg


def draw_wide(rng, cfg):
    kind = cfg["kind"]
    if kind == capi.GAME_MNK:
        m, n = int(rng.integers(3, 8)), int(rng.integers(3, 8))
        cfg.update(m=m, n=n, k=int(rng.integers(3, min(m, n, 5) + 1)), komi=0.0, enc=capi.ENC_TWOPLANE)
    elif kind == capi.GAME_C4:
        cfg.update(m=6, n=7, k=4, komi=0.0, enc=capi.ENC_TWOPLANE)
    elif kind == capi.GAME_KOMI:
        s = int(rng.integers(3, 9))
        cfg.update(m=s, n=s, k=int(rng.integers(1, 6)), komi=0.0, enc=capi.ENC_TWOPLANE)
    else:
        s = int(rng.integers(3, 10))
        cfg.update(m=s, n=s, k=0, komi=float(rng.choice([0.5, 5.5, 7.5, -3.5])), enc=capi.ENC_WQ)
    cfg["budget"] = int(rng.choice([1, 2, 7, 30, 90, 200, 320]))
    cfg["inf"] = int(rng.choice([capi.INF_HASH, capi.INF_HASH, capi.INF_DUMMY, capi.INF_UNIFORM]))
    cfg["parallel"] = int(rng.choice([1, 2, 3, 7, 11, 16]))
    cfg["a_is_black"] = tuple(int(x) for x in rng.integers(0, 2, size=int(rng.integers(1, 9))))
    cfg["DumbPass"] = bool(rng.integers(0, 2))
    cfg["PassPreference"] = int(rng.choice([capi.DONT_PREFER_PASS, capi.PREFER_PASS, capi.DONT_RESIGN]))
    cfg["ResignPercentage"] = float(rng.choice([0.0, 0.1, 0.3, 0.6, -1.0]))
    cfg["PUCT"] = float(rng.choice([1.0, 0.5, 0.25, 0.75, 0.1]))   # (0,1]: tree.go:42-44 rejects the rest
    cfg["RandomCount"] = int(rng.choice([0, 2, 5, 12]))
    cfg["RandomTemperature"] = float(rng.choice([1.0, 0.5, 0.7, 1.5, 3.0]))
    cfg["RandomMinVisits"] = int(rng.choice([0, 1, 2, 5]))
    cfg["full_game"] = bool(rng.integers(0, 3) == 0) and cfg["m"] * cfg["n"] <= 36
    return cfg


@pytest.mark.parametrize("seed", fuzz_seeds(120))
def test_random_configuration(ctx, seed):
    rng = np.random.default_rng(1000 + seed)
    c = draw_config(rng)
    policy_len = 25 if c["inf"] == capi.INF_UNIFORM else 0
    if c["inf"] == capi.INF_UNIFORM and (c["m"] * c["n"] + 1 > 25 or c["kind"] == capi.GAME_C4 and 8 > 25):
        c["inf"] = capi.INF_HASH          # the uniform example inferencer serves a 25-entry policy (5x5 boards at most)
        policy_len = 0
    run_pair(ctx, c["kind"], c["m"], c["n"], c["k"], c["komi"], enc=c["enc"], budget=c["budget"], inf=c["inf"],
             a_is_black=c["a_is_black"], max_moves=3 * c["m"] * c["n"], n_plies=0 if c["full_game"] else 14, policy_len=policy_len,
             parallel=c["parallel"], DumbPass=c["DumbPass"], PassPreference=c["PassPreference"],
             ResignPercentage=c["ResignPercentage"], PUCT=c["PUCT"], RandomCount=c["RandomCount"],
             RandomTemperature=c["RandomTemperature"], RandomMinVisits=c["RandomMinVisits"])


@pytest.mark.parametrize("seed", fuzz_seeds(12))
def test_random_large_go_board(ctx, seed):
    """wq / komi on 9..19 boards (LDS board + wave-parallel group analysis at full width), small budgets, a few plies"""
    rng = np.random.default_rng(3000 + seed)
    kind = int(rng.choice([capi.GAME_WQ, capi.GAME_WQ, capi.GAME_KOMI]))
    s = int(rng.choice([9, 11, 13, 16, 19]))
    budget = int(rng.choice([2, 6, 12]))
    lanes = int(rng.choice([1, 1, 4]))
    enc = capi.ENC_WQ if kind == capi.GAME_WQ else capi.ENC_TWOPLANE
    run_pair(ctx, kind, s, s, 0 if kind == capi.GAME_WQ else 5, 6.5 if kind == capi.GAME_WQ else 0.0, enc=enc, budget=budget,
             inf=capi.INF_HASH, a_is_black=(int(rng.integers(0, 2)), int(rng.integers(0, 2))), n_plies=int(rng.integers(4, 9)),
             parallel=lanes, DumbPass=bool(rng.integers(0, 2)), PassPreference=int(rng.choice([0, 1, 2])))
