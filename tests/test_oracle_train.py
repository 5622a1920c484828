"""CPU: the oracle's dual.Train restatement (oracle/train.hpp) — analytic backward vs central finite differences
in double, and the SGD step semantics."""
import numpy as np

import oracle_lib as O


def test_gradcheck_double():
    L = O.lib()
    assert L.orc_train_gradcheck(4, 2, 8, 3, 3, 3, 2, 10, 7, 60) < 1e-6
    assert L.orc_train_gradcheck(8, 1, 8, 4, 5, 5, 3, 26, 9, 45) < 1e-6


def test_xent_gradient_is_constant_in_logits_and_sgd_step():
    """ermahagerdmonards.go:106-147: the 'xent' is linear in the logits => d/d(Policy_b) = (1 - 2*Pi)/(B*A)."""
    B, A = 3, 10
    t = O.TrainNet(4, 1, 8, 3, 3, 2, A, B)
    t.init_random(1)
    rng = np.random.default_rng(0)
    x = rng.normal(0, 1, (B, 2, 3, 3)).astype(np.float32)
    pi = np.zeros((B, A), np.float32)
    pi[np.arange(B), [1, 4, 9]] = 1
    v = np.array([1, -1, 0], np.float32)
    names = [t.param_name(i) for i in range(t.num_params())]
    ib = names.index("Policy_b")
    before = t.get_param(ib).copy()
    t.batch(x, pi, v, lr=0.1)
    g = t.get_grad(ib).reshape(B, A)
    np.testing.assert_allclose(g, (1 - 2 * pi) / (B * A), rtol=1e-6)
    np.testing.assert_allclose(t.get_param(ib), before - 0.1 * g.ravel(), rtol=1e-6, atol=1e-8)
    # full batch-shaped learnables
    assert t.get_param(names.index("Init_gamma")).size == B * 4 * 9
    assert t.get_param(ib).size == B * A
