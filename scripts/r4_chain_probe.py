"""Round-4 probe: the chained Winograd fp16x2 block (conv_wino_h2c.hpp) against the three-kernel block.

(1) correctness on small heterogeneous nets (9x9 K=128, 19x19 K=256): chained vs three-kernel vs fp32-MFMA vs the oracle;
(2) timing at the headline shape (19x19, K=256, L=20, B=512): per kernel class (HIP events on the ctx stream, one queue) and per
    pass (two queues) for both forms.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch

torch.zeros(1, device="cuda")   # torch's HIP runtime first (as bench.py does)
import agogo_amd as A
from test_net_gpu import make_pair, rand_planes

ctx = A.Ctx(0)
out = {}

# ---- (1) correctness
for (K, L, S, B) in () if os.environ.get("PROBE_TIMING_ONLY") else ((128, 3, 9, 37), (256, 3, 19, 70), (256, 2, 19, 16), (128, 4, 19, 300), (128, 3, 7, 41)):
    onet, gnet = make_pair(ctx, K, L, 32, S, S, 18, S * S + 1, 2)
    x = rand_planes(B, 18, S, S, seed=11)
    pf, vf = gnet.infer(x)
    gnet.set_compute_mode(A.capi.COMPUTE_WINO_H2 | A.capi.COMPUTE_FORCE)
    gnet.set_wino_h2_form(0)
    p0, v0 = gnet.infer(x)
    gnet.set_wino_h2_form(int(os.environ.get("PROBE_FORM", "-1")))
    p1, v1 = gnet.infer(x)
    p1b, v1b = gnet.infer(x)
    nb = min(B, 6)
    po, vo = onet.infer(x[:nb])
    key = "K%d_L%d_S%d_B%d" % (K, L, S, B)
    out[key] = {
        "chained_vs_classic_dpol": float(np.abs(p1 - p0).max()), "chained_vs_classic_dval": float(np.abs(v1 - v0).max()),
        "chained_vs_f32_dpol": float(np.abs(p1 - pf).max()), "classic_vs_f32_dpol": float(np.abs(p0 - pf).max()),
        "chained_vs_oracle_dpol": float(np.abs(p1[:nb] - po).max()), "classic_vs_oracle_dpol": float(np.abs(p0[:nb] - po).max()),
        "chained_vs_oracle_dval": float(np.abs(v1[:nb] - vo).max()),
        "chained_repeatable": bool(np.array_equal(p1, p1b)), "identical_to_classic": bool(np.array_equal(p1, p0)),
        "finite": bool(np.all(np.isfinite(p1)) and np.all(np.isfinite(v1))), "pmax": float(p1.max()),
    }
    print(key, json.dumps(out[key]), flush=True)
    gnet.close()

# ---- (2) timing at the headline shape

S, K, L, B = 19, 256, int(os.environ.get("PROBE_L", "20")), 512
net = A.Net(ctx, K, L, 2 * K, S, S, 18, S * S + 1, bn_mode=A.capi.BN_IDENTITY)
net.init_random(1337)
for i in range(net.num_params()):
    name, n = net.param_info(i)
    if name.endswith("_gamma"):
        net.set_param(i, np.ones(n, np.float32))
    elif name.endswith("_beta"):
        net.set_param(i, np.zeros(n, np.float32))
net.commit()
net.set_compute_mode(A.capi.COMPUTE_WINO_H2)
x = torch.randint(-1, 2, (B, 18, S, S), device="cuda").float()
pol = torch.empty((B, S * S + 1), device="cuda")
val = torch.empty((B,), device="cuda")
torch.cuda.synchronize()
res = {}
for form in ((-1,) if os.environ.get("PROBE_TIMING_ONLY") else (0, 1, -1)):
    net.set_wino_h2_form(form)
    for queues in (1, 2):
        net.set_tower_queues(queues)
        for _ in range(2):
            net.infer_dev(x.data_ptr(), B, pol.data_ptr(), val.data_ptr())
        ctx.sync()
        if queues == 1:
            ctx.prof_enable(True)
        t0 = time.perf_counter()
        iters = 6
        for _ in range(iters):
            net.infer_dev(x.data_ptr(), B, pol.data_ptr(), val.data_ptr())
        ctx.sync()
        dt = (time.perf_counter() - t0) / iters
        r = {"ms_per_pass": dt * 1e3, "policy_sum": float(pol.sum().item())}
        if queues == 1:
            ctx.prof_enable(False)
            for nm, k in (("in", A.capi.PROF_WINO_IN), ("gemm", A.capi.PROF_WINO_GEMM), ("out", A.capi.PROF_WINO_OUT), ("block", A.capi.PROF_CONV)):
                n_, ms_ = ctx.prof_read(k)
                r[nm + "_ms"] = ms_ / max(n_, 1)
                r[nm + "_n"] = n_
        res["form%d_q%d" % (form, queues)] = r
        print("form", form, "queues", queues, json.dumps(r), flush=True)
out["headline"] = res
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/r4_chain_probe.json", "w"), indent=1)
