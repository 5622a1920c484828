"""Data-parallel dual.Train check (run under torch.distributed.run with 2 ranks; --shared-gpu lets both ranks use
GPU 0 with gloo on a 1-GPU box).  Each rank computes gradients on its own batch, ONE all-reduce over the flat
gradient buffer, averaged SGD step.  Rank 0 verifies against a single-process run over both batches."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import agogo_amd as A
from agogo_amd import dist as adist

ap = argparse.ArgumentParser()
ap.add_argument("--shared-gpu", action="store_true")
args = ap.parse_args()
rank, local, world = adist.init_from_env(backend="gloo" if args.shared_gpu else None)
if args.shared_gpu:
    local = 0
torch.cuda.set_device(local)
ctx = A.Ctx(local)
K, L, FC, W, H, F, Asp, B = 32, 2, 32, 5, 5, 2, 26, 4


def data(seed):
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 1, (B, F, H, W)).astype(np.float32)
    pi = np.zeros((B, Asp), np.float32); pi[np.arange(B), rng.integers(0, Asp, B)] = 1
    v = rng.choice(np.array([-1, 0, 1], np.float32), size=B).astype(np.float32)
    return x, pi, v


t = A.Trainer(ctx, K, L, FC, W, H, F, Asp, B)
t.init_random(42)                      # identical replicas
x, pi, v = data(1000 + rank)           # rank-specific batch
t.forward_backward(x, pi, v)
w = adist.allreduce_gradients(t, local)
t.apply(0.1, 1.0 / w)
ctx.sync()
ok = True
if rank == 0:
    ref = A.Trainer(ctx, K, L, FC, W, H, F, Asp, B)
    ref.init_random(42)
    gsum = None
    for r in range(world):
        ref.forward_backward(*data(1000 + r))
        g = [ref.get_grad(i) for i in range(ref.num_params())]
        gsum = g if gsum is None else [a + b for a, b in zip(gsum, g)]
    for i in range(ref.num_params()):
        want = ref.get_param(i) - 0.1 * gsum[i] / world
        got = t.get_param(i)
        scale = np.abs(want).max()
        if np.abs(got - want).max() > 1e-5 * scale + 1e-7:
            ok = False
            print("MISMATCH", ref.param_info(i)[0], np.abs(got - want).max(), scale)
    print("DP_TRAIN_CHECK", "OK" if ok else "FAIL", "world", world)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
t.close(); ctx.close()
sys.exit(0 if ok else 1)
