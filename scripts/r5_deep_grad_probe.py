"""Round-5 probe: how the device trainer's gradient error against the oracle grows with depth (K=256, 19x19, L=20, two boards), per mode
and per initialisation — is the deviation the split arithmetic's, or fp32 summation order amplified by 20 training-mode BatchNorm layers?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import agogo_amd as A
import oracle_lib as O
from agogo_amd import capi
from test_train_gpu import batch_data

ctx = A.Ctx(0)
K, L, FC, W, H, F, Aspace, B = 256, int(os.environ.get("PROBE_L", "20")), 32, 19, 19, 18, 362, 2
for wscale, beta0 in ((1.0, 0.0), (1.0, 4.0), (1.0, 8.0)):
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
        else:
            p = (p * wscale).astype(np.float32)
        ot.set_param(i, p)
        params.append(p)
    x, pi, v = batch_data(B, F, H, W, Aspace, seed=77)
    t0 = time.time()
    co = ot.batch(x, pi, v, lr=0.0)
    print("wscale", wscale, "beta0", beta0, "oracle cost", co, "oracle s %.1f" % (time.time() - t0), flush=True)
    names = [ot.param_name(i) for i in range(ot.num_params())]
    go = [ot.get_grad(i).copy() for i in range(ot.num_params())]
    for mode in ("f32", "bf16x3", "wino_h2"):
        dt = A.Trainer(ctx, K, L, FC, W, H, F, Aspace, B)
        for i, p in enumerate(params):
            dt.set_param(i, p)
        if mode != "f32":
            dt.set_compute_mode((capi.COMPUTE_BF16X3 if mode == "bf16x3" else capi.COMPUTE_WINO_H2) | capi.COMPUTE_FORCE)
        cd = dt.forward_backward(x, pi, v)
        rel = []
        frac5, frac3, l2 = [], [], []
        for i in range(len(go)):
            gd = dt.get_grad(i)
            e = np.abs(gd - go[i]); mx = float(np.abs(go[i]).max()) + 1e-30
            rel.append(float(e.max()) / mx)
            frac5.append(float((e > 2e-5 * mx).mean())); frac3.append(float((e > 1e-3 * mx).mean()))
            l2.append(float(np.sqrt((e.astype(np.float64) ** 2).sum() / ((go[i].astype(np.float64) ** 2).sum() + 1e-300))))
        stat_lines = []
        for kind in ("Filter", "_gamma", "_beta"):
            idx = [i for i in range(len(go)) if (names[i].startswith(kind) if kind == "Filter" else names[i].endswith(kind))]
            stat_lines.append("    %-7s frac>2e-5: median %.1e max %.1e | frac>1e-3: median %.1e max %.1e | rel L2: median %.1e max %.1e" % (
                kind, np.median([frac5[i] for i in idx]), max(frac5[i] for i in idx), np.median([frac3[i] for i in idx]), max(frac3[i] for i in idx),
                np.median([l2[i] for i in idx]), max(l2[i] for i in idx)))
        filt = [(names[i], rel[i]) for i in range(len(go)) if names[i].startswith("Filter")]
        print(" ", mode, "dcost %.2e" % abs(cd - co), "worst %.2e (%s)" % (max(rel), names[int(np.argmax(rel))]))
        print("    filters top->bottom:", " ".join("%.0e" % r for _, r in filt[::-1][:44:2]))
        print("\n".join(stat_lines))
        dt.close()
