"""SURVEY 8(f) row 3 measurement: example plumbing on device at the 19x19 shape (27.4 KB/example).

Times (HIP events via ctx sync + wall clock) the rotation augmenter, prepareExamples (shuffle + tensorise) and, for
comparison, what the host path costs (device->host copy of the same rows + numpy shuffle/tensorise).
Algorithmic bytes: every kernel reads each payload byte once and writes it once.
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import agogo_amd as A

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20000)
ap.add_argument("--size", type=int, default=19)
ap.add_argument("--batch", type=int, default=256)
args = ap.parse_args()
S, F, n = args.size, 18, args.n
A1 = S * S + 1
ctx = A.Ctx(0)
ex = A.Examples(ctx, F, S, S, A1)
planes = torch.randint(-1, 2, (n, F * S * S), device="cuda").float()
policy = torch.rand((n, A1), device="cuda")
value = torch.randint(-1, 2, (n,), device="cuda").float()
torch.cuda.synchronize()
row_bytes = (F * S * S + A1 + 1) * 4
ex.append_dev(planes.data_ptr(), policy.data_ptr(), value.data_ptr(), n)
ctx.sync()
t0 = time.perf_counter(); ex.augment_rotate(); ctx.sync(); t_aug = time.perf_counter() - t0
n4 = len(ex)
ex.prepare(args.batch, 0, seed=1); ctx.sync()   # warm (allocations)
t0 = time.perf_counter(); b = ex.prepare(args.batch, 0, seed=2); ctx.sync(); t_prep = time.perf_counter() - t0
# host path: what AZ.Learn does with host slices
t0 = time.perf_counter()
p, q, v = ex.get()
t_d2h = time.perf_counter() - t0
t0 = time.perf_counter()
perm = np.random.default_rng(0).permutation(n4)[: b * args.batch]
Xs, Pi, V = p[perm], q[perm], v[perm]
t_host = time.perf_counter() - t0
print(json.dumps({
    "shape": f"{S}x{S}, F={F}, {row_bytes} B/example", "examples_in": n, "after_augment": n4, "batches": b,
    "augment_ms": t_aug * 1e3, "augment_GBps": (n * row_bytes + n4 * row_bytes) / t_aug / 1e9,
    "prepare_ms": t_prep * 1e3, "prepare_GBps": 2 * b * args.batch * row_bytes / t_prep / 1e9,
    "host_path_ms": {"d2h": t_d2h * 1e3, "numpy_shuffle_tensorise": t_host * 1e3},
    "hbm_peak_GBps": 8000,
}))
