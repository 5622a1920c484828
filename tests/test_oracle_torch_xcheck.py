"""CPU: the oracle's dual-net forward against an INDEPENDENT implementation (torch.nn.functional in float64).

The reference's NN arithmetic lives in gorgonia (absent here, SURVEY 8c: "parity unpinned" against gorgonia itself).
This test pins the oracle's op semantics — cross-correlation conv with "same" zero padding, BN formula, ReLU placement,
dual-branch add, head reshapes (channel-major flatten), softmax, tanh — against a second, widely used implementation
built only from the topology in dualnet/dual.go:50-103 and ermahagerdmonards.go:33-104.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as Fn

import oracle_lib as O


def torch_forward(net, x, bn_mode, stats=None, eps=1e-5):
    c = net.conf
    K, L, FC, W, H, F, A = c["K"], c["SharedLayers"], c["FC"], c["Width"], c["Height"], c["Features"], c["ActionSpace"]
    P = [torch.from_numpy(net.get_param(i).astype(np.float64)) for i in range(net.num_params())]
    it = iter(range(len(P)))
    bi = [0]

    def conv_bn_relu(t, cin, cout, k):
        w = P[next(it)].reshape(cout, cin, k, k)
        g = P[next(it)].reshape(1, cout, H, W)
        b = P[next(it)].reshape(1, cout, H, W)
        y = Fn.conv2d(t, w, padding=k // 2)
        if bn_mode == 0:
            y = y / np.sqrt(0.0 + eps)
        elif bn_mode == 1:
            mean, var = stats[bi[0]]
            y = (y - torch.from_numpy(mean.astype(np.float64)).reshape(1, -1, 1, 1)) / torch.sqrt(
                torch.from_numpy(var.astype(np.float64)).reshape(1, -1, 1, 1) + eps)
        bi[0] += 1
        return torch.relu(y * g + b)

    t = torch.from_numpy(x.astype(np.float64))
    t = conv_bn_relu(t, F, K, 3)
    for _ in range(L):
        a = conv_bn_relu(t, K, K, 3)
        b = conv_bn_relu(t, K, K, 3)
        t = torch.relu(a + b)
    p = conv_bn_relu(t, K, 2, 1).reshape(-1, 2 * H * W)
    Wp, bp = P[next(it)].reshape(2 * H * W, A), P[next(it)]
    pol = torch.softmax(p @ Wp + bp, dim=1)
    v = conv_bn_relu(t, K, 1, 1).reshape(-1, H * W)
    W1, b1 = P[next(it)].reshape(H * W, FC), P[next(it)]
    W2, b2 = P[next(it)].reshape(FC, 1), P[next(it)]
    val = torch.tanh(torch.relu(v @ W1 + b1) @ W2 + b2).reshape(-1)
    return pol.numpy(), val.numpy()


def tame(net, seed, bn_mode):
    rng = np.random.default_rng(seed)
    for i in range(net.num_params()):
        name = net.param_name(i)
        if name.endswith("_gamma"):
            s = rng.uniform(0.5, 1.5, net.get_param(i).size).astype(np.float32)
            net.set_param(i, s * np.float32(np.sqrt(1e-5) if bn_mode == 0 else 1.0))
        elif name.endswith("_beta"):
            net.set_param(i, rng.normal(0, 0.1, net.get_param(i).size).astype(np.float32))
        elif name.endswith("_b"):
            net.set_param(i, rng.normal(0, 0.1, net.get_param(i).size).astype(np.float32))


@pytest.mark.parametrize("K,L,FC,W,H,F,A,bn_mode", [
    (3, 3, 8, 3, 3, 2, 10, 0),       # README tic-tac-toe net
    (8, 2, 16, 7, 6, 2, 8, 2),       # non-square board (Connect-4): catches H/W transposition
    (16, 2, 32, 5, 5, 18, 26, 1),    # Go-style 18 planes, running-stats BN
])
def test_oracle_forward_matches_torch_float64(K, L, FC, W, H, F, A, bn_mode):
    net = O.Net(K, L, FC, W, H, F, A, bn_mode=bn_mode)
    net.init_random(1337)
    tame(net, 7, bn_mode)
    stats = None
    if bn_mode == 1:
        rng = np.random.default_rng(3)
        stats = []
        for bidx, ch in enumerate([K] + [K, K] * L + [2, 1]):
            mean = rng.normal(0, 0.1, ch).astype(np.float32)
            var = rng.uniform(0.5, 1.5, ch).astype(np.float32)
            net.set_bn_stats(bidx, mean, var)
            stats.append((mean, var))
    rng = np.random.default_rng(11)
    x = rng.choice(np.array([-1.0, 0.0, 1.0, 0.001], dtype=np.float32), size=(5, F, H, W)).astype(np.float32)
    pol_o, val_o = net.infer(x)
    pol_t, val_t = torch_forward(net, x, bn_mode, stats)
    if K >= 8:  # (a 3-filter net can legitimately be all-dead ReLUs)
        assert np.abs(pol_t - pol_t[0]).max() > 1e-4  # boards give different outputs: not a vacuous check
    np.testing.assert_allclose(pol_o, pol_t, atol=2e-6, rtol=2e-5)
    np.testing.assert_allclose(val_o, val_t, atol=2e-6)


def test_oracle_train_forward_backward_matches_torch_autograd():
    """oracle/train.hpp (dual.Train: training-mode BN with biased batch variance, full batch-shaped gamma/beta and FC
    bias, the reference's linear 'xent' on logits + MSE on the pre-tanh value) against torch autograd in float64."""
    K, L, FC, W, H, F, A, B = 6, 2, 12, 4, 4, 3, 17, 5
    t = O.TrainNet(K, L, FC, W, H, F, A, B)
    t.init_random(5)
    rng = np.random.default_rng(2)
    names = [t.param_name(i) for i in range(t.num_params())]
    for i, nm in enumerate(names):  # make biases / beta non-trivial
        if nm.endswith("_beta") or nm.endswith("_b"):
            t.set_param(i, rng.normal(0, 0.2, t.get_param(i).size).astype(np.float32))
    x = rng.normal(0, 1, (B, F, H, W)).astype(np.float32)
    pi = np.zeros((B, A), np.float32)
    pi[np.arange(B), rng.integers(0, A, B)] = 1
    v = rng.choice(np.array([-1.0, 0.0, 1.0], np.float32), B)
    cost_o = t.batch(x, pi, v, lr=0.0)

    P = [torch.tensor(t.get_param(i).astype(np.float64), requires_grad=True) for i in range(t.num_params())]
    it = iter(range(len(P)))
    eps = 1e-5

    def conv_bn_relu(z, cin, cout, k):
        w = P[next(it)].reshape(cout, cin, k, k)
        g = P[next(it)].reshape(B, cout, H, W)
        b = P[next(it)].reshape(B, cout, H, W)
        y = Fn.conv2d(z, w, padding=k // 2)
        mean = y.mean(dim=(0, 2, 3), keepdim=True)
        var = ((y - mean) ** 2).mean(dim=(0, 2, 3), keepdim=True)
        return torch.relu((y - mean) / torch.sqrt(var + eps) * g + b)

    z = torch.tensor(x.astype(np.float64))
    z = conv_bn_relu(z, F, K, 3)
    for _ in range(L):
        a = conv_bn_relu(z, K, K, 3)
        b = conv_bn_relu(z, K, K, 3)
        z = torch.relu(a + b)
    p = conv_bn_relu(z, K, 2, 1).reshape(B, 2 * H * W)
    logits = p @ P[next(it)].reshape(2 * H * W, A) + P[next(it)].reshape(B, A)
    vv = conv_bn_relu(z, K, 1, 1).reshape(B, H * W)
    hid = torch.relu(vv @ P[next(it)].reshape(H * W, FC) + P[next(it)].reshape(B, FC))
    o = (hid @ P[next(it)].reshape(FC, 1) + P[next(it)].reshape(B, 1)).reshape(B)
    Pi, V = torch.tensor(pi.astype(np.float64)), torch.tensor(v.astype(np.float64))
    cost = -(Pi * logits + (1 - Pi) * (1 - logits)).mean() + ((o - V) ** 2).mean()
    cost.backward()
    assert abs(cost.item() - cost_o) < 1e-5 * max(1.0, abs(cost_o))
    for i, nm in enumerate(names):
        g_t = P[i].grad.numpy().ravel()
        g_o = t.get_grad(i)
        scale = max(np.abs(g_t).max(), 1e-8)
        assert np.abs(g_o - g_t).max() <= 2e-5 * scale + 1e-9, (nm, np.abs(g_o - g_t).max(), scale)
