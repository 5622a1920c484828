"""CPU: how far the ORACLE's fp32 dual.Train gradients sit from float64 autograd on one trainer-fuzz shape (tests/test_train_fuzz_gpu.py).
usage: python scripts/debug/oracle_train_vs_f64.py SEED   — prints, per data draw, the worst |g_oracle - g_f64| / max|g_f64| over the tensors.
Test infrastructure (imports oracle/ through tests/oracle_lib.py); not part of the product."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import numpy as np
import torch
import torch.nn.functional as Fn
import oracle_lib as O


def shape_of(seed):
    rng = np.random.default_rng(900 + seed)
    K = int(rng.choice([3, 8, 20, 32, 48, 64, 80, 128])); L = int(rng.integers(1, 3)); FC = int(rng.choice([2, 5, 16, 24]))
    H, W = int(rng.integers(3, 7)), int(rng.integers(3, 7)); F = int(rng.choice([1, 2, 3, 18]))
    A = int(rng.choice([3, H * W + 1, W + 1])); B = int(rng.integers(1, 8)); mode = int(rng.integers(0, 3))
    return K, L, FC, W, H, F, A, B, mode


def oracle_net(K, L, FC, W, H, F, A, B, seed=5, wscale=3.0):      # tests/test_train_gpu.py make_pair, oracle half
    ot = O.TrainNet(K, L, FC, W, H, F, A, B)
    ot.init_random(seed)
    rng = np.random.default_rng(seed)
    for i in range(ot.num_params()):
        nm, p = ot.param_name(i), ot.get_param(i)
        if nm.endswith("_gamma"):
            p = rng.uniform(0.5, 1.5, p.size).astype(np.float32)
        elif nm.endswith("_beta") or nm.endswith("_b"):
            p = rng.normal(0, 0.1, p.size).astype(np.float32)
        else:
            p = (p * wscale).astype(np.float32)
        ot.set_param(i, p)
    return ot


def f64_grads(t, x, pi, v, K, L, FC, W, H, F, A, B, eps=1e-5):
    P = [torch.tensor(t.get_param(i).astype(np.float64), requires_grad=True) for i in range(t.num_params())]
    it = iter(range(len(P)))

    def cbr(z, cin, cout, k):
        w = P[next(it)].reshape(cout, cin, k, k); g = P[next(it)].reshape(B, cout, H, W); b = P[next(it)].reshape(B, cout, H, W)
        y = Fn.conv2d(z, w, padding=k // 2)
        mean = y.mean(dim=(0, 2, 3), keepdim=True)
        var = ((y - mean) ** 2).mean(dim=(0, 2, 3), keepdim=True)
        return torch.relu((y - mean) / torch.sqrt(var + eps) * g + b)

    z = cbr(torch.tensor(x.astype(np.float64)), F, K, 3)
    for _ in range(L):
        a = cbr(z, K, K, 3); b = cbr(z, K, K, 3); z = torch.relu(a + b)
    p = cbr(z, K, 2, 1).reshape(B, 2 * H * W)
    logits = p @ P[next(it)].reshape(2 * H * W, A) + P[next(it)].reshape(B, A)
    vv = cbr(z, K, 1, 1).reshape(B, H * W)
    hid = torch.relu(vv @ P[next(it)].reshape(H * W, FC) + P[next(it)].reshape(B, FC))
    o = (hid @ P[next(it)].reshape(FC, 1) + P[next(it)].reshape(B, 1)).reshape(B)
    Pi, V = torch.tensor(pi.astype(np.float64)), torch.tensor(v.astype(np.float64))
    cost = -(Pi * logits + (1 - Pi) * (1 - logits)).mean() + ((o - V) ** 2).mean()
    cost.backward()
    return cost.item(), [q.grad.numpy().ravel() for q in P]


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
    from test_train_gpu import batch_data
    for seed in [int(a) for a in sys.argv[1:]]:
        K, L, FC, W, H, F, A, B, mode = shape_of(seed)
        ot = oracle_net(K, L, FC, W, H, F, A, B)
        for attempt in range(2):
            x, pi, v = batch_data(B, F, H, W, A, seed=seed + 1000 * attempt)
            co = ot.batch(x, pi, v, lr=0.0)
            c64, g64 = f64_grads(ot, x, pi, v, K, L, FC, W, H, F, A, B)
            worst = max((float(np.abs(ot.get_grad(i) - g64[i]).max() / max(np.abs(g64[i]).max(), 1e-30)), ot.param_name(i)) for i in range(ot.num_params()))
            print("seed %d shape %r (BN over %d values) draw %d: |cost_o - cost_f64| = %.2e, worst gradient tensor %s at %.2e of its max"
                  % (seed, (K, L, FC, W, H, F, A, B), B * H * W, attempt, abs(co - c64), worst[1], worst[0]))
