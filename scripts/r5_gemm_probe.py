"""Round-5 probe: the persistent Winograd GEMM (wino_gemm_h2p_kernel, variant 2: weight slab stationary in registers) against
wino_gemm_h2g_kernel (variant 1).

(1) correctness: policy / value of the two variants must be BIT-IDENTICAL (same MFMA sequence per accumulator) on K = 256 nets at
    batch sizes that exercise short team lists (B = 1, 3, 16), ragged last m-tiles (B = 70, 300) and 9x9 boards; vs the oracle on 6 boards;
(2) timing at the headline shape (19x19, K=256, L=20, B=512): per kernel class (HIP events on the ctx stream, one queue) and per pass
    (one and two queues) for both variants, interleaved twice.
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
ok = True

# ---- (1) correctness
for (K, L, S, B) in () if os.environ.get("PROBE_TIMING_ONLY") else ((256, 3, 19, 70), (256, 2, 19, 16), (256, 2, 19, 1), (256, 2, 19, 3), (256, 3, 19, 300), (256, 3, 9, 37), (256, 2, 9, 512)):
    onet, gnet = make_pair(ctx, K, L, 32, S, S, 18, S * S + 1, 2)
    x = rand_planes(B, 18, S, S, seed=11)
    gnet.set_compute_mode(A.capi.COMPUTE_WINO_H2 | A.capi.COMPUTE_FORCE)
    gnet.set_wino_h2_gemm(1)
    p1, v1 = gnet.infer(x)
    gnet.set_wino_h2_gemm(2)
    p2, v2 = gnet.infer(x)
    p2b, v2b = gnet.infer(x)
    nb = min(B, 6)
    po, vo = onet.infer(x[:nb])
    key = "K%d_L%d_S%d_B%d" % (K, L, S, B)
    out[key] = {
        "identical_to_h2g": bool(np.array_equal(p2, p1) and np.array_equal(v2, v1)),
        "repeatable": bool(np.array_equal(p2, p2b)),
        "max_dpol_vs_h2g": float(np.abs(p2 - p1).max()),
        "vs_oracle_dpol": float(np.abs(p2[:nb] - po).max()), "vs_oracle_dval": float(np.abs(v2[:nb] - vo).max()),
        "finite": bool(np.all(np.isfinite(p2)) and np.all(np.isfinite(v2))),
    }
    ok = ok and out[key]["identical_to_h2g"] and out[key]["repeatable"]
    print(key, json.dumps(out[key]), flush=True)
    gnet.close()
out["all_identical"] = ok

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
ref_pol = None
VARIANTS = [int(v) for v in os.environ.get("PROBE_VARIANTS", "1,2").split(",")]
QUEUES = [int(v) for v in os.environ.get("PROBE_QUEUES", "1,2").split(",")]
for rep in range(int(os.environ.get("PROBE_REPS", "2"))):
    for variant in VARIANTS:
        net.set_wino_h2_gemm(variant)
        for queues in QUEUES:
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
            pc = pol.cpu().numpy().copy()
            if ref_pol is None:
                ref_pol = pc
            r = {"ms_per_pass": dt * 1e3, "policy_identical_to_first": bool(np.array_equal(pc, ref_pol))}
            if queues == 1:
                ctx.prof_enable(False)
                for nm, k in (("in", A.capi.PROF_WINO_IN), ("gemm", A.capi.PROF_WINO_GEMM), ("out", A.capi.PROF_WINO_OUT), ("block", A.capi.PROF_CONV)):
                    n_, ms_ = ctx.prof_read(k)
                    r[nm + "_ms"] = ms_ / max(n_, 1)
                    r[nm + "_n"] = n_
            res["rep%d_gemm%d_q%d" % (rep, variant, queues)] = r
            print("rep", rep, "gemm", variant, "queues", queues, json.dumps(r), flush=True)
out["headline"] = res
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open(os.environ.get("PROBE_OUT", "gpurun_out/r5_gemm_probe.json"), "w"), indent=1)
