"""One AZ.Learn epoch (agogo.go:100-172) across N ranks, one GPU each — BASELINE configs[3]'s flow end to end:
  1. self-play: the games are sharded over the ranks (no data-path collective),
  2. the recorded examples are all-gathered (agz_examples_allgather: RCCL over xGMI; tests/fake_rccl on the 1-GPU test rig) into a device Examples set,
  3. prepareExamples with a shared seed (every rank holds the same tensors),
  4. dual.Train data-parallel: rank r takes batch (step*world + r), ONE all-reduce of the flat gradient buffer per step,
     averaged vanilla SGD (lr 0.1) — every rank ends with identical learnables,
  5. SwitchToInference (row-0 export) and the A-vs-B arena games, sharded again; wins all-reduced.
Launch: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 scripts/learn_epoch_dist.py
        (add --shared-gpu on a 1-GPU box: all ranks use GPU 0 and gloo).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

import agogo_amd as A
from agogo_amd import capi
from agogo_amd import dist as adist

ap = argparse.ArgumentParser()
ap.add_argument("--shared-gpu", action="store_true")
ap.add_argument("--size", type=int, default=3, help="board size (mnk size x size, k = size for 3, else 4)")
ap.add_argument("--K", type=int, default=32)
ap.add_argument("--L", type=int, default=2)
ap.add_argument("--games", type=int, default=64, help="self-play games in total (sharded)")
ap.add_argument("--arena-games", type=int, default=32)
ap.add_argument("--budget", type=int, default=30)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--nniters", type=int, default=2)
args = ap.parse_args()

rank, local, world = adist.init_from_env(backend="gloo" if args.shared_gpu else None)
if args.shared_gpu:
    local = 0
torch.cuda.set_device(local)
ctx = A.Ctx(local)
S, K, L = args.size, args.K, args.L
kk = 3 if S == 3 else 4
Aspace = S * S + 1
t0 = time.perf_counter()

# agents: A (current best, inference) and B (learner); identical initialisation on every rank
netA = A.Net(ctx, K, L, 2 * K, S, S, 2, Aspace)
netA.init_random(11)
netA.commit()
trainB = A.Trainer(ctx, K, L, 2 * K, S, S, 2, Aspace, args.batch)
trainB.init_random(12)

# 1. self-play, sharded
lo, hi = adist.shard_games(args.games, rank, world)
sp = A.Arena(ctx, capi.GAME_MNK, S, S, kk, encoder=capi.ENC_TWOPLANE, n_games=hi - lo, seed=1000 + lo, Budget=args.budget)
sp.set_inferencer(0, capi.INF_NET, netA)
sp.set_inferencer(1, capi.INF_NET, netA)
sp.reset()
sp.play(0, True)
t_play = time.perf_counter() - t0

# 2. + 3. gather, prepare (same seed everywhere)
ex = A.Examples(ctx, 2, S, S, Aspace)
comm = adist.make_comm(ctx) if world > 1 else A.Comm.init_all([ctx])[0]
ex.append_arena(sp)
comm.allgather_examples(ex)          # agz_examples_allgather: RCCL inside libagz
n_all = len(ex)
batches = ex.prepare(args.batch, 0, seed=77)
xd, pd, vd, rows, _ = ex.tensors_dev()
if batches < world:
    raise SystemExit("too few examples (%d) for %d ranks x batch %d" % (n_all, world, args.batch))

# 4. data-parallel dual.Train: gather rows of this rank's batch on the device, one all-reduce per step
steps = 0
cost = 0.0
xs, ps = 2 * S * S * 4, Aspace * 4     # row sizes in bytes: batches are contiguous row ranges of the prepared device tensors
n_local = batches // world
for it in range(args.nniters):
    for s in range(n_local):
        b = s * world + rank
        row0 = b * args.batch
        last = it == args.nniters - 1 and s == n_local - 1
        # agz_trainer_forward_backward_allreduce_dev: the gradient slices are summed over the ranks under the backward pass
        c = comm.forward_backward_allreduce_dev(trainB, xd + row0 * xs, pd + row0 * ps, vd + row0 * 4, want_cost=last)
        if last:
            cost = c
        trainB.apply(0.1, 1.0 / comm.size())
        steps += 1
ctx.sync()
# every rank must hold the same learnables
digest = torch.tensor([float(np.abs(trainB.get_param(i)).sum()) for i in range(trainB.num_params())], dtype=torch.float64)
same = True
if world > 1:
    ds = [torch.zeros_like(digest) for _ in range(world)]
    dist.all_gather(ds, digest)
    same = all(torch.equal(d, ds[0]) for d in ds)

# 5. SwitchToInference + arena games A vs B, sharded
netB = A.Net(ctx, K, L, 2 * K, S, S, 2, Aspace)
trainB.export(netB)
alo, ahi = adist.shard_games(args.arena_games, rank, world)
ev = A.Arena(ctx, capi.GAME_MNK, S, S, kk, encoder=capi.ENC_TWOPLANE, n_games=max(ahi - alo, 1), seed=5000 + alo, Budget=args.budget)
ev.set_inferencer(0, capi.INF_NET, netA)
ev.set_inferencer(1, capi.INF_NET, netB)
ev.reset()
ev.play(0, False)
r = ev.results()
wins = torch.tensor([r["a_wins"], r["b_wins"], r["draws"]], dtype=torch.float64)
if world > 1:
    dist.all_reduce(wins)
if rank == 0:
    print(json.dumps({"LEARN_EPOCH": "OK" if same else "REPLICAS DIVERGED", "world": world, "selfplay_games": args.games,
                      "examples_gathered": n_all, "batches": batches, "dp_steps_per_rank": steps, "last_cost": cost,
                      "replicas_identical": same, "arena": {"a_wins": int(wins[0]), "b_wins": int(wins[1]), "draws": int(wins[2])},
                      "seconds": time.perf_counter() - t0, "selfplay_seconds": t_play}))
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
for h in (comm, ev, netB, ex, sp, trainB, netA):
    h.close()
ctx.close()
sys.exit(0 if same else 1)
