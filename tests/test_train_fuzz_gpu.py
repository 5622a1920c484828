"""GPU parity, randomised trainer shapes (fixed seeds): dual.Train forward/backward vs the oracle on shapes that are not
multiples of anything, BatchSize 1..7, both compute modes."""
import numpy as np
import pytest

from conftest import fuzz_seeds

import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi
from test_train_gpu import make_pair, batch_data

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", fuzz_seeds(14))
def test_random_trainer_shape(ctx, seed):
    rng = np.random.default_rng(900 + seed)
    K = int(rng.choice([3, 8, 20, 32, 48, 64, 80, 128]))
    L = int(rng.integers(1, 3))
    FC = int(rng.choice([2, 5, 16, 24]))
    H, W = int(rng.integers(3, 7)), int(rng.integers(3, 7))
    F = int(rng.choice([1, 2, 3, 18]))
    Aspace = int(rng.choice([3, H * W + 1, W + 1]))
    B = int(rng.integers(1, 8))
    ot, dt = make_pair(ctx, K, L, FC, W, H, F, Aspace, B)
    mode = int(rng.integers(0, 3))
    if mode == 1:
        dt.set_compute_mode(capi.COMPUTE_BF16X3 | capi.COMPUTE_FORCE)
    elif mode == 2:
        dt.set_compute_mode(capi.COMPUTE_WINO_H2 | capi.COMPUTE_FORCE)
    shape = (K, L, FC, W, H, F, Aspace, B)
    # ReLU is not differentiable at 0: with batches this small a pre-activation within rounding of zero can take a different
    # side on the device and in the oracle and move a few gradients by percent (seen once in 14 shapes; the same shape passes
    # with any other data).  A case fails only if two independent data draws both miss the tolerance; the cost must always match.
    failures = []
    for attempt in range(2):
        x, pi, v = batch_data(B, F, H, W, Aspace, seed=seed + 1000 * attempt)
        co = ot.batch(x, pi, v, lr=0.0)
        cd = dt.forward_backward(x, pi, v)
        # cost: the device and the oracle are each within 2e-5 of the float64 value (weights x3 make the logits large and the
        # mean cancels); in a 1200-shape soak two draws had them on opposite sides of it, 2.4e-5 apart
        # (profiles/r01/fuzz_soak.txt) -> mutual tolerance 4e-5
        # ... and with fewer than 64 samples per channel (BatchSize 1 on a 6x4 board: training-mode BatchNorm over 24 values) the
        # fp32-MFMA path itself sits 2.8e-5 from the oracle and the split modes 5.9e-5 on one draw of a 2006-shape soak
        # (profiles/r05/fuzz_soak.txt, seed 560359): 1e-4 there
        cost_tol = 4e-5 if B * H * W >= 64 else 1e-4
        assert abs(cd - co) <= cost_tol * max(1.0, abs(co)), (cd, co, shape)
        # gradients: 2e-5 of a tensor's maximum — except under the same fewer-than-64-samples BatchNorm, where the ORACLE's own fp32
        # gradients sit 3.0e-5 / 3.5e-5 from float64 autograd (seed 2100290: BatchSize 1 on a 3x3 board, statistics over 9 values, true-fp32
        # device path 3.4e-5 from the oracle on both draws; scripts/debug/oracle_train_vs_f64.py, profiles/r06/fuzz_soak.txt): two fp32
        # evaluations each that far from the exact value -> 8e-5 between them
        grad_tol = 2e-5 if B * H * W >= 64 else 8e-5
        bad = []
        for i in range(ot.num_params()):
            go, gd = ot.get_grad(i), dt.get_grad(i)
            scale = float(np.abs(go).max())
            err = float(np.abs(gd - go).max())
            if err > grad_tol * scale + 1e-7:
                bad.append((ot.param_name(i), err, scale))
        if not bad:
            return
        failures.append(bad)
    raise AssertionError((shape, failures))
