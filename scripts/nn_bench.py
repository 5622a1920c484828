"""NN-only micro-benchmark of the conv tower on the BASELINE 19x19 config (K=256, L=20, B=512)."""
import argparse
import json
import sys
import os
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import agogo_amd as A

ap = argparse.ArgumentParser()
ap.add_argument("--K", type=int, default=256)
ap.add_argument("--L", type=int, default=20)
ap.add_argument("--B", type=int, default=512)
ap.add_argument("--size", type=int, default=19)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--x3", action="store_true")
ap.add_argument("--h2", action="store_true", help="AGZ_COMPUTE_FP16X2")
ap.add_argument("--wino", action="store_true", help="AGZ_COMPUTE_WINO (AGZ_WINO_CHUNK=n in the environment: boards per chunk)")
ap.add_argument("--wino-h2", action="store_true", help="AGZ_COMPUTE_WINO_H2")
ap.add_argument("--force", action="store_true", help="AGZ_COMPUTE_FORCE: take the split kernels below the chip-filling threshold")
ap.add_argument("--no-latency", action="store_true", help="agz_net_set_latency_mode(0)")
ap.add_argument("--host", action="store_true", help="also time the host-buffer boundary agz_net_infer (pageable and agz_host_alloc buffers): the PCIe-inclusive rate")
ap.add_argument("--zero", action="store_true", help="all-zero weights (DVFS probe: same instruction stream, low toggle power)")
args = ap.parse_args()
ctx = A.Ctx(0)
S = args.size
net = A.Net(ctx, args.K, args.L, 2 * args.K, S, S, 18, S * S + 1)
if not args.zero:
    net.init_random(1337)
net.commit()
if args.x3:
    net.set_compute_mode(A.capi.COMPUTE_BF16X3)
if args.h2:
    net.set_compute_mode(A.capi.COMPUTE_FP16X2)
if args.wino:
    net.set_compute_mode(A.capi.COMPUTE_WINO)
if args.wino_h2:
    net.set_compute_mode(A.capi.COMPUTE_WINO_H2 | (A.capi.COMPUTE_FORCE if args.force else 0))
if args.no_latency:
    net.set_latency_mode(False)
x = torch.randint(-1, 2, (args.B, 18, S, S), device="cuda").float()
pol = torch.empty((args.B, S * S + 1), device="cuda")
val = torch.empty((args.B,), device="cuda")
torch.cuda.synchronize()
for _ in range(2):
    net.infer_dev(x.data_ptr(), args.B, pol.data_ptr(), val.data_ptr())
ctx.sync()
ctx.prof_enable(True)
t0 = time.perf_counter()
for _ in range(args.iters):
    net.infer_dev(x.data_ptr(), args.B, pol.data_ptr(), val.data_ptr())
ctx.sync()
t1 = time.perf_counter()
ctx.prof_enable(False)
n_conv, ms_conv = ctx.prof_read(A.capi.PROF_CONV)
n_head, ms_head = ctx.prof_read(A.capi.PROF_HEADS)
n_init, ms_init = ctx.prof_read(A.capi.PROF_CONV_INIT)
wino = {}
if args.wino or args.wino_h2:
    for nm, k in (("in", A.capi.PROF_WINO_IN), ("gemm", A.capi.PROF_WINO_GEMM), ("out", A.capi.PROF_WINO_OUT)):
        n_, ms_ = ctx.prof_read(k)
        wino[nm + "_ms_avg"] = ms_ / max(n_, 1)
        wino[nm + "_launches"] = n_
flops = net.flops_per_eval() * args.B
dt = (t1 - t0) / args.iters
host = {}
if args.host:
    from agogo_amd.capi import lib, _pf, _check
    xh0 = x.cpu().numpy()
    for kind in ("pageable", "pinned"):
        mk = (lambda shp: np.zeros(shp, np.float32)) if kind == "pageable" else (lambda shp: ctx.host_array(shp))
        xh = mk((args.B, 18, S, S)); xh[...] = xh0
        ph = mk((args.B, S * S + 1)); vh = mk((args.B,))
        for _ in range(2):
            _check(lib().agz_net_infer(net.h, _pf(xh), args.B, _pf(ph), _pf(vh)), "agz_net_infer")
        th = time.perf_counter()
        for _ in range(args.iters):
            _check(lib().agz_net_infer(net.h, _pf(xh), args.B, _pf(ph), _pf(vh)), "agz_net_infer")
        dh = (time.perf_counter() - th) / args.iters
        host[kind] = {"ms_per_pass": dh * 1e3, "evals_per_s": args.B / dh, "bytes_in": int(xh.nbytes), "bytes_out": int(ph.nbytes + vh.nbytes),
                      "policy_equal_to_device_path": bool(np.array_equal(ph, pol.cpu().numpy()))}
print(json.dumps({"B": args.B, "K": args.K, "L": args.L, "ms_per_pass": dt * 1e3, "evals_per_s": args.B / dt,
                  "tflops": flops / dt / 1e12, "frac_fp32_peak": flops / dt / 157.3e12,
                  "conv_launches": n_conv, "conv_ms_avg": ms_conv / max(n_conv, 1), "heads_ms_avg": ms_head / max(n_head, 1), "init_ms_avg": ms_init / max(n_init, 1),
                  "wino": wino, "host_boundary": host, "policy_sum": float(pol.sum().item()), "value_mean": float(val.mean().item())}))
