"""GPU parity, randomised network shapes (fixed seeds): channel counts that are not multiples of the tile sizes, non-square
boards, odd feature counts, all BatchNorm readings, every compute mode (forced below the chip-filling threshold), batch
sizes on both sides of the latency regime — libagz inference vs the oracle, the stated fp32 tolerance."""
import numpy as np
import pytest

from conftest import fuzz_seeds

import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi
from test_net_gpu import make_pair, rand_planes, POL_ATOL, POL_RTOL, VAL_ATOL

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", fuzz_seeds(28))
def test_random_network_shape(ctx, seed):
    rng = np.random.default_rng(500 + seed)
    K = int(rng.choice([3, 8, 20, 32, 48, 64, 96, 128, 192]))
    L = int(rng.integers(0, 4))
    FC = int(rng.choice([2, 7, 16, 33, 64]))
    H, W = int(rng.integers(3, 10)), int(rng.integers(3, 10))
    F = int(rng.choice([1, 2, 3, 18, 32]))
    Aspace = int(rng.choice([3, H * W + 1, W + 1, 2 * H * W]))
    Aspace = min(max(Aspace, 3), 512)
    bn_mode = int(rng.integers(0, 3))
    B = int(rng.choice([1, 2, 5, 17, 40]))
    mode = int(rng.choice([capi.COMPUTE_F32_MFMA, capi.COMPUTE_BF16X3, capi.COMPUTE_FP16X2]))
    onet, gnet = make_pair(ctx, K, L, FC, W, H, F, Aspace, bn_mode, seed=seed + 7)
    gnet.set_compute_mode(mode | capi.COMPUTE_FORCE)
    if rng.integers(0, 2):
        gnet.set_latency_mode(False)
    x = rand_planes(B, F, H, W, seed=seed)
    pol_g, val_g = gnet.infer(x)
    pol_o, val_o = onet.infer(x)
    assert np.all(np.isfinite(pol_g)) and np.all(np.isfinite(val_g))
    np.testing.assert_allclose(pol_g.sum(axis=1), 1.0, atol=1e-5)
    np.testing.assert_allclose(pol_g, pol_o, atol=POL_ATOL, rtol=POL_RTOL, err_msg=str((K, L, FC, W, H, F, Aspace, bn_mode, B, mode)))
    np.testing.assert_allclose(val_g, val_o, atol=VAL_ATOL)


@pytest.mark.parametrize("wmode", [capi.COMPUTE_WINO, capi.COMPUTE_WINO_H2], ids=["wino", "wino_h2"])
@pytest.mark.parametrize("seed", fuzz_seeds(16))
def test_random_network_shape_winograd(ctx, seed, wmode):
    """the same family in the Winograd modes (forced below the chip-filling threshold): K a multiple of 64 (with and without the
    C % 128 fast paths of WINO_H2: lane-swapped stores, the board range reduced by the next input transform), boards from 3x3 (one
    ragged tile) to 13x13 — WINO_H2 picks F(5x5,3x3) or F(4x4,3x3) by row count —, batches on both sides of the latency regime"""
    rng = np.random.default_rng(2500 + seed)
    K = int(rng.choice([64, 128, 192, 256]))
    L = int(rng.integers(1, 4))
    FC = int(rng.choice([2, 7, 16, 33, 64]))
    H, W = int(rng.integers(3, 14)), int(rng.integers(3, 14))
    F = int(rng.choice([1, 2, 3, 18]))
    Aspace = min(max(int(rng.choice([3, H * W + 1, W + 1])), 3), 512)
    bn_mode = int(rng.integers(0, 3))
    B = int(rng.choice([3, 9, 33, 70]))
    onet, gnet = make_pair(ctx, K, L, FC, W, H, F, Aspace, bn_mode, seed=seed + 11)
    gnet.set_compute_mode(wmode | capi.COMPUTE_FORCE)
    if rng.integers(0, 2):
        gnet.set_latency_mode(False)
    x = rand_planes(B, F, H, W, seed=seed)
    pol_g, val_g = gnet.infer(x)
    idx = sorted(set([0, B // 2, B - 1]))
    pol_o, val_o = onet.infer(x[idx])
    assert np.all(np.isfinite(pol_g)) and np.all(np.isfinite(val_g))
    np.testing.assert_allclose(pol_g.sum(axis=1), 1.0, atol=1e-5)
    np.testing.assert_allclose(pol_g[idx], pol_o, atol=POL_ATOL, rtol=POL_RTOL, err_msg=str((K, L, FC, W, H, F, Aspace, bn_mode, B)))
    np.testing.assert_allclose(val_g[idx], val_o, atol=VAL_ATOL)
