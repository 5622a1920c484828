"""GPU parity: libagz dual-net forward (HIP, fp32 MFMA) vs the CPU oracle restatement.

Tolerance: fp32 — |policy - oracle| <= 2e-5 absolute (+1e-4 relative), |value - oracle| <= 1e-4.
Both sides compute in fp32; they differ only in summation order (MFMA k-pair order vs sequential),
BN folding (scale/shift vs normalise-then-affine) and expf/tanhf implementations.
"""
import numpy as np
import pytest

import agogo_amd as A
import oracle_lib as O

pytestmark = pytest.mark.gpu

POL_ATOL, POL_RTOL, VAL_ATOL = 2e-5, 1e-4, 1e-4


def make_pair(ctx, K, L, FC, W, H, F, Aspace, bn_mode=0, seed=1337, tame=True):
    """tame=True rescales BN gamma so the folded per-(c,h,w) scale is O(1): with the raw reference
    initialiser kinds the degenerate-eps BN (x316 per layer) saturates softmax/tanh and every board
    gives the same one-hot policy — a parity check on that alone would be vacuous."""
    onet = O.Net(K, L, FC, W, H, F, Aspace, bn_mode=bn_mode)
    onet.init_random(seed)
    if tame:
        rng = np.random.default_rng(seed + 1)
        for i in range(onet.num_params()):
            name = onet.param_name(i)
            if name.endswith("_gamma"):
                s = rng.uniform(0.5, 1.5, onet.get_param(i).size).astype(np.float32)
                onet.set_param(i, s * np.float32(np.sqrt(1e-5) if bn_mode == 0 else 1.0))
            elif name.endswith("_beta"):
                onet.set_param(i, rng.normal(0, 0.1, onet.get_param(i).size).astype(np.float32))
    gnet = A.Net(ctx, K, L, FC, W, H, F, Aspace, bn_mode=bn_mode)
    assert gnet.num_params() == onet.num_params()
    for i in range(onet.num_params()):
        name, n = gnet.param_info(i)
        p = onet.get_param(i)
        assert n == p.size, (name, n, p.size)
        gnet.set_param(i, p)
    if bn_mode == A.capi.BN_RUNNING:
        rng = np.random.default_rng(seed)
        chans = [K] + [K, K] * L + [2, 1]
        for bi, c in enumerate(chans):
            mean = rng.normal(0, 0.1, c).astype(np.float32)
            var = rng.uniform(0.5, 1.5, c).astype(np.float32)
            onet.set_bn_stats(bi, mean, var)
            gnet.set_bn_stats(bi, mean, var)
    gnet.commit()
    return onet, gnet


def rand_planes(B, F, H, W, seed):
    rng = np.random.default_rng(seed)
    x = rng.choice(np.array([-1.0, 0.0, 1.0, 0.001], dtype=np.float32), size=(B, F, H, W))
    return x.astype(np.float32)


CASES = [
    # K, L, FC, W, H, F, A, B, bn_mode
    (3, 3, 8, 3, 3, 2, 10, 7, 0),        # TTT config #1 (README.md:90-109): K=3 -> padded 32, cfg1 kernel
    (64, 2, 128, 7, 6, 2, 8, 5, 0),      # C4-shaped, cfg0 kernel, partial M tile
    (32, 2, 64, 9, 9, 18, 82, 4, 0),     # 9x9 Go, K=32 (cfg1)
    (128, 2, 256, 9, 9, 18, 82, 3, 0),   # 9x9 Go K=128 (config #3 width)
    (64, 1, 128, 19, 19, 18, 362, 3, 2),  # 19x19, identity BN
    (64, 1, 128, 19, 19, 18, 362, 2, 1),  # 19x19, running-stats BN
    (256, 2, 256, 19, 19, 18, 362, 1, 2),  # batch 1, K=256: tournament Agent.Search shape -> split-K path (18 splits)
    (128, 1, 128, 19, 19, 18, 362, 2, 0),  # batch 2, split-K with 128-column tiles
    (64, 1, 64, 9, 9, 18, 82, 200, 2),     # 16200 rows: enough tiles that split-K is NOT taken
]


def test_infer_matches_oracle_reference_init(ctx):
    """raw reference initialiser kinds + degenerate-eps BN (saturating, but parity must still hold)."""
    onet, gnet = make_pair(ctx, 32, 2, 64, 5, 5, 2, 26, bn_mode=0, tame=False)
    x = rand_planes(6, 2, 5, 5, seed=11)
    pol_o, val_o = onet.infer(x)
    pol_g, val_g = gnet.infer(x)
    np.testing.assert_allclose(pol_g, pol_o, atol=1e-4, rtol=1e-3)
    np.testing.assert_allclose(val_g, val_o, atol=1e-4)


@pytest.mark.parametrize("K,L,FC,W,H,F,Aspace,B,bn_mode", CASES)
def test_infer_matches_oracle(ctx, K, L, FC, W, H, F, Aspace, B, bn_mode):
    onet, gnet = make_pair(ctx, K, L, FC, W, H, F, Aspace, bn_mode)
    x = rand_planes(B, F, H, W, seed=K * 100 + B)
    pol_o, val_o = onet.infer(x)
    pol_g, val_g = gnet.infer(x)
    assert np.all(np.isfinite(pol_g)) and np.all(np.isfinite(val_g))
    np.testing.assert_allclose(pol_g.sum(axis=1), 1.0, atol=1e-5)
    np.testing.assert_allclose(pol_g, pol_o, atol=POL_ATOL, rtol=POL_RTOL)
    np.testing.assert_allclose(val_g, val_o, atol=VAL_ATOL)
    # not a degenerate comparison: outputs differ across boards (a 3-filter net can legitimately be all-dead ReLUs)
    if K >= 8 and B > 1:
        assert np.abs(pol_o - pol_o[0]).max() > 1e-6


def test_batch_independence(ctx):
    """meta.go:175-189: every board is its own row-0 evaluation — results must not depend on batch composition.
    Bitwise within a regime (agz_net_set_latency_mode): 130 boards run the throughput tower, 1..3 boards the split-K one."""
    onet, gnet = make_pair(ctx, 64, 2, 128, 9, 9, 18, 82)
    x = rand_planes(130, 18, 9, 9, seed=5)  # spans two 128-row M tiles per board group
    pol_all, val_all = gnet.infer(x)
    pol_1, val_1 = gnet.infer(x[77:78])
    pol_3, val_3 = gnet.infer(x[76:79])
    # same (latency) regime: bit-identical whatever the batch
    np.testing.assert_array_equal(pol_3[1], pol_1[0])
    np.testing.assert_array_equal(val_3[1], val_1[0])
    # across regimes: fp32 summation order differs, nothing else
    np.testing.assert_allclose(pol_all[77], pol_1[0], atol=POL_ATOL, rtol=POL_RTOL)
    np.testing.assert_allclose(val_all[77], val_1[0], atol=VAL_ATOL)
    # latency regime off: strict bitwise independence at every batch size
    gnet.set_latency_mode(False)
    pol_1s, val_1s = gnet.infer(x[77:78])
    np.testing.assert_array_equal(pol_all[77], pol_1s[0])
    np.testing.assert_array_equal(val_all[77], val_1s[0])
    pol_o, val_o = onet.infer(x[77:78])
    np.testing.assert_allclose(pol_1s, pol_o, atol=POL_ATOL, rtol=POL_RTOL)
    np.testing.assert_allclose(pol_1, pol_o, atol=POL_ATOL, rtol=POL_RTOL)


def test_asymmetric_weights_transpose_detect(ctx):
    """one-hot filters with asymmetric taps: catches row/col or tap-order swaps exactly."""
    K, H, W, F = 64, 5, 5, 2
    gnet = A.Net(ctx, K, 0, 8, W, H, F, 26, bn_mode=2)
    onet = O.Net(K, 0, 8, W, H, F, 26, bn_mode=2)
    rng = np.random.default_rng(3)
    for i in range(onet.num_params()):
        p = onet.get_param(i)
        name = onet.param_name(i)
        if name == "FilterInit":
            p = np.zeros((K, F, 3, 3), np.float32)
            for o in range(K):
                p[o, o % F, (o // F) % 3, (o // (3 * F)) % 3] = 1.0 + o  # asymmetric one-hot taps
        elif name.endswith("_gamma"):
            p = np.ones_like(p)
        elif name.endswith("_beta"):
            p = np.zeros_like(p)
        else:
            p = rng.normal(0, 0.1, p.size).astype(np.float32)
        onet.set_param(i, p)
        gnet.set_param(i, p)
    gnet.commit()
    x = rng.normal(0, 1, (3, F, H, W)).astype(np.float32)
    pol_o, val_o = onet.infer(x)
    pol_g, val_g = gnet.infer(x)
    np.testing.assert_allclose(pol_g, pol_o, atol=POL_ATOL, rtol=POL_RTOL)
    np.testing.assert_allclose(val_g, val_o, atol=VAL_ATOL)


def test_init_random_matches_oracle_rng(ctx):
    """agz_net_init_random and the oracle state the same SplitMix64/Glorot recipe independently."""
    onet = O.Net(32, 1, 64, 5, 5, 2, 26)
    onet.init_random(99)
    gnet = A.Net(ctx, 32, 1, 64, 5, 5, 2, 26)
    gnet.init_random(99)
    for i in range(onet.num_params()):
        np.testing.assert_array_equal(gnet.get_param(i), onet.get_param(i))


def test_infer_before_commit_fails(ctx):
    gnet = A.Net(ctx, 32, 1, 64, 5, 5, 2, 26)
    with pytest.raises(A.AgzError):
        gnet.infer(np.zeros((1, 2, 5, 5), np.float32))


def test_checkpoint_roundtrip(ctx, tmp_path):
    """AZ.Save / Load analogue (agogo.go:175-209; dualnet TestEncodeDecode dual_test.go:111-141): learnables equal
    after decode, outputs identical, mismatched configuration rejected."""
    onet, gnet = make_pair(ctx, 32, 2, 64, 5, 5, 2, 26, bn_mode=1)
    path = str(tmp_path / "example.model")
    gnet.save(path)
    other = A.Net(ctx, 32, 2, 64, 5, 5, 2, 26, bn_mode=1)
    other.load(path)
    for i in range(gnet.num_params()):
        np.testing.assert_array_equal(other.get_param(i), gnet.get_param(i))
    x = rand_planes(3, 2, 5, 5, seed=4)
    p0, v0 = gnet.infer(x)
    p1, v1 = other.infer(x)
    np.testing.assert_array_equal(p0, p1)
    np.testing.assert_array_equal(v0, v1)
    wrong = A.Net(ctx, 32, 1, 64, 5, 5, 2, 26, bn_mode=1)
    with pytest.raises(A.AgzError):
        wrong.load(path)


SPLIT_MODES = [A.capi.COMPUTE_BF16X3, A.capi.COMPUTE_FP16X2]


@pytest.mark.parametrize("mode", SPLIT_MODES)
@pytest.mark.parametrize("K,L,FC,W,H,F,Aspace,B,bn_mode", [
    (64, 2, 128, 9, 9, 18, 82, 70, 0),      # 5670 rows, partial last M tile
    (128, 2, 64, 9, 9, 18, 82, 33, 2),      # two column tiles
    (256, 2, 128, 19, 19, 18, 362, 12, 2),  # BASELINE width (12 boards: just above the K>=256 latency-regime bound)
    (64, 5, 64, 9, 9, 18, 82, 64, 1),       # deeper tower, running-stats BN
    (192, 1, 64, 7, 6, 2, 8, 37, 2),        # K=192: three column tiles (fp16x2 takes its narrow kernel), 6x7 board, ragged M
    (128, 1, 32, 5, 5, 2, 26, 90, 2),       # 5x5 board, 2250 rows: tiles straddle many boards, last tile partial
])
def test_split_compute_modes_match_oracle_and_f32(ctx, K, L, FC, W, H, F, Aspace, B, bn_mode, mode):
    """AGZ_COMPUTE_BF16X3 (conv_x3.hpp: exact 3-way bf16 split, 6 of 9 piece products) and AGZ_COMPUTE_FP16X2
    (conv_h2.hpp: power-of-two scaling + 2-way fp16 split, 3 of 4 products, device-tracked activation ranges): the same
    fp32 tolerance against the oracle as the fp32-MFMA path, and agreement with that path far inside it."""
    onet, gnet = make_pair(ctx, K, L, FC, W, H, F, Aspace, bn_mode)
    x = rand_planes(B, F, H, W, seed=K + B)
    pol_f, val_f = gnet.infer(x)
    gnet.set_compute_mode(mode | A.capi.COMPUTE_FORCE)   # small test shapes: below the chip-filling threshold
    pol_g, val_g = gnet.infer(x)
    gnet.set_compute_mode(A.capi.COMPUTE_F32_MFMA)
    assert not np.array_equal(pol_g, pol_f)          # really a different arithmetic path
    idx = [0, B // 2, B - 1]
    pol_o, val_o = onet.infer(x[idx])
    np.testing.assert_allclose(pol_g[idx], pol_o, atol=POL_ATOL, rtol=POL_RTOL)
    np.testing.assert_allclose(val_g[idx], val_o, atol=VAL_ATOL)
    np.testing.assert_allclose(pol_g, pol_f, atol=POL_ATOL, rtol=POL_RTOL)
    np.testing.assert_allclose(val_g, val_f, atol=VAL_ATOL)
    print("mode %d max |dpol| vs f32: %.3e  vs oracle: %.3e   f32 vs oracle: %.3e" % (
        mode, np.abs(pol_g - pol_f).max(), np.abs(pol_g[idx] - pol_o).max(), np.abs(pol_f[idx] - pol_o).max()))


@pytest.mark.parametrize("mode", SPLIT_MODES)
@pytest.mark.parametrize("in_scale", [1e-4, 1.0, 3e3])
def test_split_compute_modes_over_input_and_activation_ranges(ctx, mode, in_scale):
    """Range stress: inputs scaled by 1e-4 .. 3e3 and the reference's raw initialiser kinds under the degenerate-eps
    BatchNorm (activations grow ~x316 per layer).  fp16x2 derives its per-layer scale from the tracked
    maximum, so neither overflow nor flush-to-zero may appear; compared with the fp32-MFMA path on the pre-softmax side
    of saturation: the probabilities must agree to the usual tolerance."""
    onet, gnet = make_pair(ctx, 64, 3, 64, 9, 9, 18, 82, bn_mode=0, tame=False)
    x = rand_planes(40, 18, 9, 9, seed=9) * np.float32(in_scale)
    pol_f, val_f = gnet.infer(x)
    gnet.set_compute_mode(mode | A.capi.COMPUTE_FORCE)   # small test shapes: below the chip-filling threshold
    pol_g, val_g = gnet.infer(x)
    assert np.all(np.isfinite(pol_g)) and np.all(np.isfinite(val_g))
    np.testing.assert_allclose(pol_g.sum(axis=1), 1.0, atol=1e-5)
    np.testing.assert_allclose(pol_g, pol_f, atol=1e-4, rtol=1e-3)
    np.testing.assert_allclose(val_g, val_f, atol=1e-4)
