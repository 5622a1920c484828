"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" on CPU).

Self-play shards by independent games: zero data-path collectives.  The only exchange steps are the gather of
per-rank example buffers before dual.Train (agogo.go:118-133) and the gradient sum of the data-parallel step — both RCCL calls
INSIDE libagz (agz_examples_allgather, agz_trainer_allreduce); this module only sets the process group up, ships the
communicator id and aggregates the bench counters.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None, timeout_s=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher (torch.distributed.run).  timeout_s: the process group's collective
    time-out (default: torch's own)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # both device types: host-side bookkeeping tensors (digests, win counts) go over gloo, device buffers over RCCL
            # (an nccl-only group raises "No backend type associated with device type cpu" on a CPU tensor: ADVICE r1)
            backend = "cpu:gloo,cuda:nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if timeout_s:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, local, world


def exchange_unique_id(group=None):
    """rank 0 draws the RCCL unique id (agz_comm_unique_id), the process group ships its 128 bytes (host side, any backend)"""
    from . import capi
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [capi.Comm.unique_id() if rank == 0 else None]
    if world > 1:
        dist.broadcast_object_list(box, src=0, group=group)
    return box[0]


def make_comm(ctx, group=None):
    """An agz_comm (RCCL inside libagz, include/agz.h) for this rank's ctx: rank 0 draws the RCCL unique id, the process
    group ships it, every rank joins with agz_comm_init_rank.  The exchange itself then runs entirely inside libagz:
    Comm.allgather_examples / Comm.allreduce_trainer — what a Go host calls through cgo.  (There is no torch-side copy of the
    gather or of the gradient sum any more: one implementation, the one the boundary exports.)"""
    from . import capi
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    return capi.Comm.init_rank(ctx, world, rank, exchange_unique_id(group))


def shard_games(total_games, rank, world):
    """games [lo, hi) owned by `rank` (config #4: 4096 games sharded 512/GPU)"""
    per = total_games // world
    extra = total_games % world
    lo = rank * per + min(rank, extra)
    hi = lo + per + (1 if rank < extra else 0)
    return lo, hi


def reduce_step_timing(seconds, counts, group=None, device=None):
    """bench.py's aggregation over ranks: MAX of the timed region's wall time, SUM of the per-rank counters (sims, evals, ...).
    Returns (t_max, [sums...]).  One rank: the inputs."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return float(seconds), [float(c) for c in counts]
    tt = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=group)
    cc = torch.tensor([float(c) for c in counts], dtype=torch.float64, device=device)
    dist.all_reduce(cc, op=dist.ReduceOp.SUM, group=group)
    return float(tt.item()), [float(x) for x in cc.tolist()]
