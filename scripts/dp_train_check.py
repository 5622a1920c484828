"""Data-parallel dual.Train check (run under torch.distributed.run with 2 ranks; --shared-gpu lets both ranks use
GPU 0 on a 1-GPU box: gloo for the process group, AGZ_RCCL_LIB=tests/fake_rccl/librccl_fake.so for libagz's collectives).  Each rank computes gradients on its own batch, ONE all-reduce over the flat
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
comm = adist.make_comm(ctx) if world > 1 else A.Comm.init_all([ctx])[0]
t.forward_backward(x, pi, v)
comm.allreduce_trainer(t)          # ONE RCCL all-reduce inside libagz over the flat gradient buffer
w = comm.size()
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

# ---- phase 2: the per-epoch example exchange on the device: arena buffers -> all-gather -> Examples ----
from agogo_amd import capi


def play(seed):
    a = A.Arena(ctx, capi.GAME_MNK, 3, 3, 3, encoder=capi.ENC_TWOPLANE, n_games=6, seed=seed, Budget=20)
    a.set_inferencer(0, capi.INF_HASH)
    a.set_inferencer(1, capi.INF_HASH)
    a.reset()
    a.play(0, True)
    return a


mine = play(500 + rank)
ex = A.Examples(ctx, 2, 3, 3, 10)
ex.append_arena(mine)              # canonical (episode) order first, then the exchange inside libagz
comm.allgather_examples(ex)
n_all = len(ex)
got = ex.get()
batches = ex.prepare(8, 0, seed=31)       # same seed on every rank -> identical training tensors everywhere
X, P, V = ex.tensors()
digest = torch.tensor([float(np.abs(X).sum() + 3 * np.abs(P).sum() + 7 * V.sum()), float(n_all), float(batches)], dtype=torch.float64)
if world > 1:
    ds = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(ds, digest)
    same = all(torch.equal(d, ds[0]) for d in ds)
else:
    same = True
ok2 = same
if rank == 0:
    want = [[], [], []]
    for r in range(world):                 # every rank's games are reproducible from its seed
        a = play(500 + r)
        pl, po, va, gi = a.examples()
        o = np.argsort(gi, kind="stable")
        for k, arr in enumerate((pl[o], po[o], va[o])):
            want[k].append(arr)
    for k in range(3):
        w_ = np.concatenate(want[k], axis=0)
        if w_.shape != got[k].shape or not np.array_equal(w_, got[k]):
            ok2 = False
            print("GATHER MISMATCH", k, w_.shape, got[k].shape)
    print("EXAMPLE_GATHER_CHECK", "OK" if ok2 else "FAIL", "world", world, "examples", n_all, "batches", batches)
ok = ok and ok2
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
comm.close(); ex.close(); mine.close(); t.close(); ctx.close()
sys.exit(0 if ok else 1)
