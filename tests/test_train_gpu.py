"""GPU parity: dual.Train on device (agz_trainer_*) vs the oracle restatement (oracle/train.hpp).

Tolerance (fp32 both sides, different summation orders, float atomics in the weight gradient): gradients
|d - o| <= 2e-5 * max|o| + 1e-7 per tensor; cost 1e-5 relative; parameters after three SGD steps |d - o| <= 1e-4 * max|o| per tensor.
The oracle's backward itself is pinned by a finite-difference check in double (tests/test_oracle_train.py)."""
import numpy as np
import pytest

import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi

pytestmark = pytest.mark.gpu


def make_pair(ctx, K, L, FC, W, H, F, Aspace, B, seed=5, wscale=3.0):
    ot = O.TrainNet(K, L, FC, W, H, F, Aspace, B)
    ot.init_random(seed)
    rng = np.random.default_rng(seed)
    for i in range(ot.num_params()):
        nm = ot.param_name(i)
        p = ot.get_param(i)
        if nm.endswith("_gamma"):
            p = rng.uniform(0.5, 1.5, p.size).astype(np.float32)
        elif nm.endswith("_beta") or nm.endswith("_b"):
            p = rng.normal(0, 0.1, p.size).astype(np.float32)
        else:
            p = (p * wscale).astype(np.float32)
        ot.set_param(i, p)
    dt = A.Trainer(ctx, K, L, FC, W, H, F, Aspace, B)
    assert dt.num_params() == ot.num_params()
    for i in range(ot.num_params()):
        name, n = dt.param_info(i)
        assert n == ot.get_param(i).size, (name, ot.param_name(i))
        dt.set_param(i, ot.get_param(i))
        np.testing.assert_array_equal(dt.get_param(i), ot.get_param(i))  # layout round trip
    return ot, dt


def batch_data(B, F, H, W, Aspace, seed):
    rng = np.random.default_rng(seed)
    x = rng.choice(np.array([-1.0, 0.0, 1.0, 0.001], np.float32), size=(B, F, H, W)).astype(np.float32)
    pi = np.zeros((B, Aspace), np.float32)
    pi[np.arange(B), rng.integers(0, Aspace, B)] = 1.0
    v = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), size=B).astype(np.float32)
    return x, pi, v


CASES = [
    # K, L, FC, W, H, F, A, B
    (32, 1, 16, 3, 3, 2, 10, 4),      # tic-tac-toe shaped
    (32, 2, 64, 5, 5, 2, 26, 6),
    (64, 2, 64, 7, 6, 2, 8, 5),       # connect-4 shaped, K=64 (cfg0 conv kernels), non-square
    (128, 1, 64, 9, 9, 18, 82, 3),    # 9x9 go shaped
    (3, 3, 8, 3, 3, 2, 10, 5),        # the README tic-tac-toe net (DefaultConf(3,3,10), K=3 SharedLayers=3): K padded 3 -> 32
    (20, 1, 8, 4, 4, 2, 17, 1),       # BatchSize 1 (batch statistics over 16 pixels only), K=20 padded
    (40, 2, 24, 5, 4, 3, 21, 7),      # nothing a multiple of anything
    (64, 1, 16, 16, 17, 3, 273, 2),   # a board >= 16 wide that is not 19x19 (non-square, 272 = 8.5 K steps of 32 rows per board): the
                                      # three-tap DMA weight gradient with a 64-channel operand in a 128-column tile (masked lanes)
]


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "wino_h2"])
@pytest.mark.parametrize("K,L,FC,W,H,F,Aspace,B", CASES)
def test_forward_backward_matches_oracle(ctx, K, L, FC, W, H, F, Aspace, B, mode):
    ot, dt = make_pair(ctx, K, L, FC, W, H, F, Aspace, B)
    if mode == "bf16x3":   # bf16x3 forward / data-gradient convolutions (Cin >= 64) and weight gradient; forced below the chip-filling threshold
        dt.set_compute_mode(capi.COMPUTE_BF16X3 | capi.COMPUTE_FORCE)
    elif mode == "wino_h2":   # Winograd fp16x2 forward / data-gradient convolutions (Cin >= 64, weights transformed on the device)
        dt.set_compute_mode(capi.COMPUTE_WINO_H2 | capi.COMPUTE_FORCE)
    x, pi, v = batch_data(B, F, H, W, Aspace, seed=K + B)
    co = ot.batch(x, pi, v, lr=0.0)
    cd = dt.forward_backward(x, pi, v)
    assert abs(cd - co) <= 1e-5 * max(1.0, abs(co)), (cd, co)
    for i in range(ot.num_params()):
        go, gd = ot.get_grad(i), dt.get_grad(i)
        scale = float(np.abs(go).max())
        err = float(np.abs(gd - go).max())
        assert err <= 2e-5 * scale + 1e-7, (ot.param_name(i), err, scale)
    # the gradient is not trivially zero
    assert any(np.abs(ot.get_grad(i)).max() > 1e-6 for i in range(ot.num_params()))


@pytest.mark.parametrize("K,L,FC,W,H,F,Aspace,B", CASES)
def test_first_form_of_the_head_kernels_still_matches_the_oracle(ctx, K, L, FC, W, H, F, Aspace, B):
    """Round 5 replaced the nine head kernels (one thread per output, three-block BatchNorm passes: 1.85 ms of a G19 step) by a second form
    (wave-per-row reductions, multi-block partial sums, FC kernels on LDS tiles), the default every other test runs; the first form stays
    reachable (agz_trainer_set_dma_forward bit 2, agz_debug.h) and is held to the same bar here."""
    ot, dt = make_pair(ctx, K, L, FC, W, H, F, Aspace, B)
    dt.set_dma_forward(1 | 4)
    x, pi, v = batch_data(B, F, H, W, Aspace, seed=K + B)
    co = ot.batch(x, pi, v, lr=0.0)
    cd = dt.forward_backward(x, pi, v)
    assert abs(cd - co) <= 1e-5 * max(1.0, abs(co)), (cd, co)
    for i in range(ot.num_params()):
        go, gd = ot.get_grad(i), dt.get_grad(i)
        scale = float(np.abs(go).max())
        assert float(np.abs(gd - go).max()) <= 2e-5 * scale + 1e-7, ot.param_name(i)


def test_forward_dma_convolution_forms_agree(ctx):
    """k_conv_h2dma3 (nine taps of a 32-channel chunk from one x image, 256 x 128 tile; default) against k_conv_h2dma (one tap per K step,
    128 x 256 tile; agz_trainer_set_dma_forward bit 5): the same products in the same order per accumulator, so the forward pass — the cost —
    is identical, and the gradients agree to the weight gradient's atomics.  K=256 / 19x19, three boards: 1083 rows = four whole 256-row
    tiles and a partial one, board crossings inside tiles 1 and 2; two dual blocks."""
    K, L, FC, W, H, F, Aspace, B = 256, 2, 32, 19, 19, 18, 362, 3
    ot, d3 = make_pair(ctx, K, L, FC, W, H, F, Aspace, B)
    d1 = A.Trainer(ctx, K, L, FC, W, H, F, Aspace, B)
    for i in range(ot.num_params()):
        d1.set_param(i, ot.get_param(i))
    for dt in (d3, d1):
        dt.set_compute_mode(capi.COMPUTE_WINO_H2 | capi.COMPUTE_FORCE)
    d1.set_dma_forward(1 | 32)
    for seed in (5, 6):
        x, pi, v = batch_data(B, F, H, W, Aspace, seed=seed)
        c3 = d3.forward_backward(x, pi, v)
        c1 = d1.forward_backward(x, pi, v)
        assert c3 == c1, (seed, c3, c1)
        for i in range(ot.num_params()):
            g3, g1 = d3.get_grad(i), d1.get_grad(i)
            scale = float(np.abs(g1).max())
            assert float(np.abs(g3 - g1).max()) <= 2e-6 * scale + 1e-9, (seed, ot.param_name(i))


@pytest.mark.parametrize("mode", ["bf16x3", "wino_h2", "wino_h2_staged_fwd", "wino_h2_weights_in_line", "wino_h2_one_stream", "wino_h2_one_tap_per_step"])
def test_forward_backward_headline_width_19x19(ctx, mode):
    """the trainer's fast modes at the headline tower's width and board (K=256, 19x19, 18 planes, 362 actions; one dual block,
    two boards so the oracle finishes in seconds): the shapes the G19 step runs — 128-multiple tiles, F(5x5,3x3) with the ragged
    last tile row and column, 512-channel data gradient — against the oracle.

    Both modes: the strict per-tensor tolerance of the small shapes (every gradient tensor within 2e-5 of its maximum) on EVERY one
    of several data draws.  (Round 2's Winograd FORWARD convolution did not meet this — a ReLU unit flipped against the oracle on
    every other draw — and was removed from the trainer; AGZ_COMPUTE_WINO_H2 now puts only the data gradient on the Winograd path.)"""
    K, L, FC, W, H, F, Aspace, B = 256, 1, 32, 19, 19, 18, 362, 2
    ot, dt = make_pair(ctx, K, L, FC, W, H, F, Aspace, B)
    dt.set_compute_mode((capi.COMPUTE_BF16X3 if mode == "bf16x3" else capi.COMPUTE_WINO_H2) | capi.COMPUTE_FORCE)
    if mode == "wino_h2_staged_fwd":     # round 5: the forward convolutions default to the DMA GEMM on pre-split planes (k_conv_h2dma); this keeps
        dt.set_dma_forward(False)        # the staging-split kernel (conv3x3_h2w_kernel) under the same bar
    if mode == "wino_h2_weights_in_line":  # ... and every layer's weight images are built at the start of the step on the side stream
        dt.set_dma_forward(1 | 8)          # (prep_weights, train.hip); bit 3 keeps the per-layer in-line passes under the same bar
    if mode == "wino_h2_one_tap_per_step":  # the first form of the DMA forward convolution (k_conv_h2dma; the default is k_conv_h2dma3)
        dt.set_dma_forward(1 | 32)
    if mode == "wino_h2_one_stream":       # diagnostic form (bit 4): no side stream, every kernel on the step's stream
        dt.set_dma_forward(1 | 16)
    report = []
    for seed in (77, 78, 79, 80):
        x, pi, v = batch_data(B, F, H, W, Aspace, seed=seed)
        co = ot.batch(x, pi, v, lr=0.0)
        cd = dt.forward_backward(x, pi, v)
        # (the cost is a mean of large cancelling logit terms — weights x3, 362 actions — so its error does not scale with |cost|:
        #  3e-4 absolute in every mode incl. fp32-MFMA on these draws; the small-shape tests hold it to 1e-5 relative)
        assert abs(cd - co) <= 1e-4 * max(1.0, abs(co)), (seed, cd, co)
        worst = 0.0
        for i in range(ot.num_params()):
            go, gd = ot.get_grad(i), dt.get_grad(i)
            scale = float(np.abs(go).max())
            err = float(np.abs(gd - go).max())
            worst = max(worst, err / (scale + 1e-30))
            assert err <= 2e-5 * scale + 1e-7, (seed, ot.param_name(i), err, scale)
        report.append("draw %d: worst %.1e" % (seed, worst))
    # the same draw once more: the layer inputs' ranges now equal last step's exactly, so the fp16 planes the DMA convolution and the weight
    # gradient read are the ones k_bn_apply_v wrote under the estimate (k_split_h2p_cond keeps them) — same bar
    cd2 = dt.forward_backward(x, pi, v)
    assert abs(cd2 - co) <= 1e-4 * max(1.0, abs(co))
    for i in range(ot.num_params()):
        go, gd = ot.get_grad(i), dt.get_grad(i)
        scale = float(np.abs(go).max())
        assert float(np.abs(gd - go).max()) <= 2e-5 * scale + 1e-7, ("repeat", ot.param_name(i))
    print("trainer %s at K=256 / 19x19 (strict tolerance 2e-5 of the tensor maximum): %s" % (mode, "; ".join(report)))


def test_fused_gamma_beta_step_equals_the_two_pass_step(ctx):
    """agz_trainer_batch lets the BatchNorm backward kernel take the SGD step of the batch-shaped gamma / beta in place (98 % of the
    learnables: no gradient round trip); agz_trainer_forward_backward + agz_trainer_apply (the data-parallel path) materialises the
    gradients and sweeps.  Same expression per element: the parameters after a step agree to the weight-gradient atomics' noise."""
    K, L, FC, W, H, F, Aspace, B = 64, 2, 32, 7, 7, 2, 50, 6
    _, t1 = make_pair(ctx, K, L, FC, W, H, F, Aspace, B, seed=21)
    _, t2 = make_pair(ctx, K, L, FC, W, H, F, Aspace, B, seed=21)
    for step in range(2):
        x, pi, v = batch_data(B, F, H, W, Aspace, seed=300 + step)
        c1 = t1.batch(x, pi, v, lr=0.1)                 # fused
        c2 = t2.forward_backward(x, pi, v)              # two passes
        t2.apply(0.1)
        assert abs(c1 - c2) <= 1e-6 * max(1.0, abs(c2))
    for i in range(t1.num_params()):
        a, b = t1.get_param(i), t2.get_param(i)
        name = t1.param_info(i)[0]
        if name.endswith("_gamma") or name.endswith("_beta"):
            np.testing.assert_allclose(a, b, rtol=0, atol=2e-6 * max(float(np.abs(b).max()), 1e-3), err_msg=name)
        else:
            assert float(np.abs(a - b).max()) <= 2e-6 * max(float(np.abs(b).max()), 1e-3), name


def test_sgd_steps_and_export(ctx):
    """three dual.Train inner-loop steps (meta.go:33-40, lr 0.1), then dual.Infer's row-0 copy into an inference net."""
    K, L, FC, W, H, F, Aspace, B = 32, 2, 32, 5, 5, 2, 26, 4
    ot, dt = make_pair(ctx, K, L, FC, W, H, F, Aspace, B, seed=9)
    costs = []
    for step in range(3):
        x, pi, v = batch_data(B, F, H, W, Aspace, seed=100 + step)
        co = ot.batch(x, pi, v, lr=0.1)
        cd = dt.batch(x, pi, v, lr=0.1)
        assert abs(cd - co) <= 2e-5 * max(1.0, abs(co))
        costs.append(cd)
    for i in range(ot.num_params()):
        po, pd = ot.get_param(i), dt.get_param(i)
        scale = float(np.abs(po).max())
        assert float(np.abs(pd - po).max()) <= 1e-4 * scale + 1e-7, ot.param_name(i)
    # export: inference on row-0 parameters equals the oracle inference net built from row 0
    net = A.Net(ctx, K, L, FC, W, H, F, Aspace, bn_mode=capi.BN_DEGENERATE_EPS)
    dt.export(net)
    onet = O.Net(K, L, FC, W, H, F, Aspace, bn_mode=0)
    for i in range(onet.num_params()):
        full = ot.get_param(i)
        onet.set_param(i, full[: onet.get_param(i).size])
        scale = float(np.abs(onet.get_param(i)).max())
        assert float(np.abs(net.get_param(i) - onet.get_param(i)).max()) <= 1e-4 * scale + 1e-7
    x = batch_data(3, F, H, W, Aspace, seed=7)[0]
    pg, vg = net.infer(x)
    assert np.all(np.isfinite(pg)) and np.all(np.isfinite(vg))
    # ... and the exported network COMPUTES what the oracle's does.  Under the degenerate-eps reading every BatchNorm multiplies by 316 and
    # the heads saturate (nothing to compare), and three lr-0.1 steps on the pair's 3x-Glorot filters blow the head weights up to ~40: the
    # comparison runs on a second pair with Glorot-scale filters, under IDENTITY statistics (gamma * x + beta with the TRAINED row-0
    # gamma / beta).  The two parameter sets agree to 1e-4 (three SGD steps on two arithmetic paths), the outputs accordingly.
    ot2, dt2 = make_pair(ctx, K, L, FC, W, H, F, Aspace, B, seed=9, wscale=1.0)
    for step in range(3):
        xb, pib, vb = batch_data(B, F, H, W, Aspace, seed=100 + step)
        ot2.batch(xb, pib, vb, lr=0.1)
        dt2.batch(xb, pib, vb, lr=0.1)
    net2 = A.Net(ctx, K, L, FC, W, H, F, Aspace, bn_mode=capi.BN_IDENTITY)
    dt2.export(net2)
    onet2 = O.Net(K, L, FC, W, H, F, Aspace, bn_mode=2)
    for i in range(onet2.num_params()):
        onet2.set_param(i, ot2.get_param(i)[: onet2.get_param(i).size])
    pg2, vg2 = net2.infer(x)
    po2, vo2 = onet2.infer(x)
    assert 1.5 / Aspace < float(po2.max()) < 0.9 and 1e-3 < float(np.abs(vo2).max()) < 0.9, (float(po2.max()), float(np.abs(vo2).max()))   # not saturated
    np.testing.assert_allclose(pg2, po2, atol=2e-5, rtol=1e-3)
    np.testing.assert_allclose(vg2, vo2, atol=1e-4)


_DEEP = {}


def _deep_oracle(beta0):
    """the oracle's gradients of ONE batch at the headline tower's depth (K=256, 19x19, L=20, two boards: ~20 s of oracle time per
    regime), shared by the modes.  beta0 is added to every BatchNorm beta: 4.0 keeps every ReLU unit active (see the test)."""
    if beta0 not in _DEEP:
        K, L, FC, W, H, F, Aspace, B = 256, 20, 32, 19, 19, 18, 362, 2
        ot = O.TrainNet(K, L, FC, W, H, F, Aspace, B)
        ot.init_random(5)
        rng = np.random.default_rng(5)
        params = []
        for i in range(ot.num_params()):
            nm = ot.param_name(i)
            p = ot.get_param(i)
            if nm.endswith("_gamma"):
                p = rng.uniform(0.5, 1.5, p.size).astype(np.float32)
            elif nm.endswith("_beta") or nm.endswith("_b"):
                p = (rng.normal(0, 0.1, p.size) + (beta0 if nm.endswith("_beta") else 0.0)).astype(np.float32)
            ot.set_param(i, p)
            params.append(p)
        x, pi, v = batch_data(B, F, H, W, Aspace, seed=77)
        cost = ot.batch(x, pi, v, lr=0.0)
        _DEEP[beta0] = dict(shape=(K, L, FC, W, H, F, Aspace, B), params=params, names=[ot.param_name(i) for i in range(ot.num_params())],
                            grads=[ot.get_grad(i).copy() for i in range(ot.num_params())], data=(x, pi, v), cost=cost)
    return _DEEP[beta0]


def _deep_errors(ctx, D, mode):
    K, L, FC, W, H, F, Aspace, B = D["shape"]
    dt = A.Trainer(ctx, K, L, FC, W, H, F, Aspace, B)
    for i, p in enumerate(D["params"]):
        dt.set_param(i, p)
    if mode != "f32":
        dt.set_compute_mode((capi.COMPUTE_BF16X3 if mode == "bf16x3" else capi.COMPUTE_WINO_H2) | capi.COMPUTE_FORCE)
    cd = dt.forward_backward(*D["data"])
    rel, l2 = [], []
    for i, go in enumerate(D["grads"]):
        e = np.abs(dt.get_grad(i) - go).astype(np.float64)
        rel.append(float(e.max()) / (float(np.abs(go).max()) + 1e-30))
        l2.append(float(np.sqrt((e ** 2).sum() / ((go.astype(np.float64) ** 2).sum() + 1e-300))))
    dt.close()
    return cd, rel, l2


def test_forward_backward_headline_depth_l20(ctx):
    """the trainer's measured arithmetic (AGZ_COMPUTE_WINO_H2: fp16x2 forward convolutions, Winograd fp16x2 data
    gradient, fp16x2 three-tap weight gradient; and bf16x3) at the depth train_leg times — K=256, 19x19, TWENTY dual blocks — against the
    oracle trainer, all 129 gradient tensors, so that error compounding through 20 layers of backward is observed, not assumed.

    What was observed (scripts/r5_deep_grad_probe.py, profiles/r05/deep_gradient_probe.log): with ordinary BatchNorm shifts the
    comparison at this depth is decided by ReLU units whose pre-activation rounds to the other side of zero — a handful among 7.4 M
    units, in EVERY mode including the true-fp32 kernels: every tensor then differs coherently by ~1e-2 (relative L2: fp32-MFMA 1.2-1.7e-2,
    bf16x3 5e-3) although the cost agrees to 1e-4.  That is the function's discontinuity, not arithmetic.  The arithmetic's own compounding
    is measured where the function is smooth: beta + 4 keeps every unit active (the tower is then 20 blocks of convolution and training-
    mode BatchNorm, forward and backward).  There fp32-MFMA — summation order alone — reaches 2.4-2.7e-5 of a tensor's maximum at depth
    20 (2e-5 is the bar of the one-block test) and the split modes stay within twice that.  Bars: 5e-5 per tensor in every mode, and
    no split mode worse than 2x the fp32-MFMA path's own worst tensor.  The ordinary-shift regime is reported, with a gross-error bar."""
    D = _deep_oracle(4.0)
    worst = {}
    for mode in ("f32", "bf16x3", "wino_h2"):
        cd, rel, l2 = _deep_errors(ctx, D, mode)
        assert abs(cd - D["cost"]) <= 1e-4 * max(1.0, abs(D["cost"])), (mode, cd, D["cost"])
        w = int(np.argmax(rel))
        worst[mode] = rel[w]
        print("trainer %s at K=256 / 19x19 / L=20, every unit active: worst gradient tensor %s at %.2e of its maximum; relative L2 median %.1e max %.1e"
              % (mode, D["names"][w], rel[w], float(np.median(l2)), max(l2)))
        for i, r in enumerate(rel):
            assert r <= 5e-5, (mode, D["names"][i], r)
    assert worst["bf16x3"] <= 2.0 * worst["f32"] and worst["wino_h2"] <= 2.0 * worst["f32"], worst
    Dn = _deep_oracle(0.0)
    for mode in ("f32", "wino_h2"):
        cd, rel, l2 = _deep_errors(ctx, Dn, mode)
        assert abs(cd - Dn["cost"]) <= 1e-3 * max(1.0, abs(Dn["cost"])), (mode, cd, Dn["cost"])
        print("trainer %s at L=20, ordinary shifts (ReLU flips decide): relative L2 per tensor median %.1e max %.1e, worst element %.1e of its tensor's maximum"
              % (mode, float(np.median(l2)), max(l2), max(rel)))
        assert max(l2) <= 0.15, (mode, max(l2))      # gross-error bar only: see the docstring


def test_trained_weights_through_the_measured_inference_arithmetic(ctx):
    """the inference modes had only ever seen Glorot weights.  Three agz_trainer_batch steps (lr 0.1) at K=128 / 9x9 /
    four blocks, then export -> commit (which proves the range bound g1 * max|x| + g0 of DESIGN 4a on THESE weights) -> inference under
    AGZ_COMPUTE_WINO_H2 (chained blocks), bf16x3 and fp32-MFMA against the oracle inference net holding the same exported row-0
    parameters: the full network tolerance."""
    from test_net_gpu import POL_ATOL, POL_RTOL, VAL_ATOL
    K, L, FC, W, H, F, Aspace, B = 128, 4, 64, 9, 9, 18, 82, 8
    ot = O.TrainNet(K, L, FC, W, H, F, Aspace, B)
    ot.init_random(13)
    rng = np.random.default_rng(13)
    start = []
    # (gamma ~ U(0.3, 0.9): with the reference's N(0, sigma) gamma a four-block tower's output is constant, with gamma ~ 1 three lr-0.1 steps
    #  at this width saturate the softmax under identity statistics — explored with the oracle alone; here the policy peaks at ~5x uniform)
    for i in range(ot.num_params()):
        nm, p = ot.param_name(i), ot.get_param(i)
        if nm.endswith("_gamma"):
            p = (0.6 * rng.uniform(0.5, 1.5, p.size)).astype(np.float32)
        elif nm.endswith("_beta") or nm.endswith("_b"):
            p = rng.normal(0, 0.1, p.size).astype(np.float32)
        ot.set_param(i, p)
        start.append(p.copy())
    dt = A.Trainer(ctx, K, L, FC, W, H, F, Aspace, B)
    for i in range(ot.num_params()):
        dt.set_param(i, ot.get_param(i))
    dt.set_compute_mode(capi.COMPUTE_WINO_H2 | capi.COMPUTE_FORCE)
    for step in range(3):
        x, pi, v = batch_data(B, F, H, W, Aspace, seed=500 + step)
        co = ot.batch(x, pi, v, lr=0.1)
        cd = dt.batch(x, pi, v, lr=0.1)
        assert abs(cd - co) <= 2e-3 * max(1.0, abs(co)), (step, cd, co)
    moved = 0.0
    # (648 samples per channel and 1.3 M ReLU units: a unit whose pre-activation rounds to the other side of zero moves a filter gradient
    #  by ~1 % — see test_forward_backward_headline_depth_l20 — so the two trainers' parameters are only held to a gross-error bar here;
    #  the strict trainer parity lives in the tests above, this test is about the INFERENCE arithmetic on weights SGD has moved)
    for i in range(ot.num_params()):
        po, pd = ot.get_param(i), dt.get_param(i)
        assert float(np.abs(pd - po).max()) <= 5e-2 * float(np.abs(po).max()) + 1e-7, ot.param_name(i)
    net = A.Net(ctx, K, L, FC, W, H, F, Aspace, bn_mode=capi.BN_IDENTITY)
    dt.export(net)                                         # (export commits)
    onet = O.Net(K, L, FC, W, H, F, Aspace, bn_mode=2)
    for i in range(onet.num_params()):
        onet.set_param(i, net.get_param(i))                # the SAME exported parameters on both sides
        moved = max(moved, float(np.abs(net.get_param(i) - start[i][: net.get_param(i).size]).max()))
    assert moved > 1e-3                                    # SGD really moved them
    xs = batch_data(40, F, H, W, Aspace, seed=9)[0]
    po, vo = onet.infer(xs[:6])
    assert 2.0 / Aspace < float(po.max()) < 0.95 and float(np.abs(vo).max()) < 0.95, (float(po.max()), float(np.abs(vo).max()))   # a real comparison
    for mode in (capi.COMPUTE_F32_MFMA, capi.COMPUTE_BF16X3 | capi.COMPUTE_FORCE, capi.COMPUTE_WINO_H2 | capi.COMPUTE_FORCE):
        net.set_compute_mode(mode)
        pg, vg = net.infer(xs)
        np.testing.assert_allclose(pg[:6], po, atol=POL_ATOL, rtol=POL_RTOL, err_msg="mode %d" % mode)
        np.testing.assert_allclose(vg[:6], vo, atol=VAL_ATOL, err_msg="mode %d" % mode)
        print("trained weights, mode %d: max |dpolicy| vs oracle %.2e" % (mode, float(np.abs(pg[:6] - po).max())))
    net.close()
    dt.close()


def test_agz_train_loop_runs_and_shuffles(ctx):
    """dual.Train(d, Xs, policies, values, batches, iterations): iterations x batches steps + shuffleBatch."""
    K, L, FC, W, H, F, Aspace, B = 32, 1, 16, 3, 3, 2, 10, 8
    dt = A.Trainer(ctx, K, L, FC, W, H, F, Aspace, B)
    dt.init_random(3)
    batches = 3
    x, pi, v = batch_data(B * batches, F, H, W, Aspace, seed=1)
    x0 = x.copy()
    p_before = dt.get_param(0).copy()
    cost = dt.train(x.reshape(B * batches, -1), pi, v, batches, 2, seed=11)
    assert np.isfinite(cost)
    assert not np.array_equal(dt.get_param(0), p_before)
    assert not np.array_equal(x, x0) and np.array_equal(np.sort(x.reshape(B * batches, -1), axis=0), np.sort(x0.reshape(B * batches, -1), axis=0))


def test_data_parallel_step_two_ranks(ctx):
    """C2: two ranks, rank-specific batches, ONE all-reduce over the flat gradient buffer, averaged SGD step
    == single-process average.  Both ranks share GPU 0 here (process group over gloo, libagz's collectives through
    tests/fake_rccl); on a multi-GPU node the same code runs over RCCL."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", AGZ_RCCL_LIB=os.path.join(root, "tests", "fake_rccl", "librccl_fake.so"))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29577", os.path.join(root, "scripts", "dp_train_check.py"), "--shared-gpu"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert "DP_TRAIN_CHECK OK world 2" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]
    # phase 2 of the script: arena example buffers -> all-gather -> device Examples set, identical on both ranks
    assert "EXAMPLE_GATHER_CHECK OK world 2" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_trainer_checkpoint_resume(ctx, tmp_path):
    """save after one step, load into a fresh trainer, continue: identical to the uninterrupted run (bitwise up to the
    weight-gradient atomics: 1e-6 relative), and a mismatched configuration is refused."""
    K, L, FC, W, H, F, Aspace, B = 32, 1, 16, 3, 3, 2, 10, 4
    t1 = A.Trainer(ctx, K, L, FC, W, H, F, Aspace, B)
    t1.init_random(5)
    x, pi, v = batch_data(B, F, H, W, Aspace, seed=1)
    x2, pi2, v2 = batch_data(B, F, H, W, Aspace, seed=2)
    t1.batch(x, pi, v, 0.1)
    path = tmp_path / "trainer.agz"
    t1.save(path)
    t1.batch(x2, pi2, v2, 0.1)
    t2 = A.Trainer(ctx, K, L, FC, W, H, F, Aspace, B)
    t2.load(path)
    t2.batch(x2, pi2, v2, 0.1)
    for i in range(t1.num_params()):
        a, b = t1.get_param(i), t2.get_param(i)
        assert np.abs(a - b).max() <= 1e-6 * max(np.abs(a).max(), 1e-3), t1.param_info(i)
    other = A.Trainer(ctx, K, L, FC, W, H, F, Aspace, B + 1)
    with pytest.raises(A.AgzError, match="not a checkpoint of this trainer"):
        other.load(path)
    with pytest.raises(A.AgzError):
        t2.load(tmp_path / "missing.agz")


def test_trainer_init_random_is_the_oracles_sequential_stream(ctx):
    """agz_trainer_init_random generates the oracle's SplitMix64/Box-Muller stream in parallel chunks (counter-based RNG):
    bit-identical to the sequential restatement, including tensors large enough for the threaded path."""
    K, L, FC, W, H, F, Aspace, B = 64, 1, 32, 9, 9, 18, 82, 16    # gamma/beta: 16*64*81 = 82,944 floats per tensor
    ot = O.TrainNet(K, L, FC, W, H, F, Aspace, B)
    ot.init_random(4242)
    dt = A.Trainer(ctx, K, L, FC, W, H, F, Aspace, B)
    dt.init_random(4242)
    for i in range(ot.num_params()):
        np.testing.assert_array_equal(dt.get_param(i).view(np.uint32), ot.get_param(i).view(np.uint32), err_msg=ot.param_name(i))


def test_learn_epoch_two_ranks_end_to_end(ctx):
    """scripts/learn_epoch_dist.py: sharded self-play -> example all-gather -> shared-seed prepareExamples -> data-parallel
    dual.Train (one gradient all-reduce per step) -> SwitchToInference -> sharded arena games; two ranks on GPU 0 (tests/fake_rccl).
    The replicas must end with identical learnables."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", AGZ_RCCL_LIB=os.path.join(root, "tests", "fake_rccl", "librccl_fake.so"))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", "29591", os.path.join(root, "scripts", "learn_epoch_dist.py"), "--shared-gpu"],
                         capture_output=True, text=True, timeout=300, env=env)
    line = [l for l in out.stdout.splitlines() if l.startswith("{") and "LEARN_EPOCH" in l]
    assert line, out.stdout[-2000:] + out.stderr[-2000:]
    r = json.loads(line[-1])
    assert r["LEARN_EPOCH"] == "OK" and r["world"] == 2 and r["replicas_identical"]
    assert r["examples_gathered"] > 64 and r["dp_steps_per_rank"] >= 2
    assert r["arena"]["a_wins"] + r["arena"]["b_wins"] + r["arena"]["draws"] == 32
