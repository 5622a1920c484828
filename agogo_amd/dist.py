"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" == RCCL on ROCm, "gloo" on CPU).

Self-play shards by independent games: zero data-path collectives.  The only exchange step is the
gather of per-rank example buffers before dual.Train (agogo.go:118-133): variable-count all-gather below.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the launcher (torch.distributed.run)."""
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
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


class _DevArray:
    """zero-copy view of a libagz device buffer for torch.as_tensor (CUDA array interface v2)"""

    def __init__(self, ptr, shape, typestr="<f4"):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def device_tensor(ptr, shape, device):
    return torch.as_tensor(_DevArray(ptr, shape), device=device)


def all_gather_examples(planes, policy, value, group=None):
    """Variable-count all-gather of example records {Board[F*HW], Policy[A+1], Value} (datatypes.go:41-46).

    Inputs are this rank's [n_r, *] tensors (any device the backend supports).  Returns the concatenation
    over ranks in rank order.  Direct all-gather of padded blocks: with RCCL every peer pair uses its own
    xGMI link concurrently, so the exchange is bound by one link's bandwidth per peer, not by a ring.
    """
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return planes, policy, value
    world = dist.get_world_size(group)
    n = torch.tensor([planes.shape[0]], dtype=torch.int64, device=planes.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    nmax = max(counts)
    if nmax == 0:
        return planes, policy, value
    out = []
    for t in (planes, policy, value.reshape(-1, 1)):
        pad = torch.zeros((nmax, t.shape[1]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(bufs, pad, group=group)
        out.append(torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0))
    return out[0], out[1], out[2].reshape(-1)


def gather_into_examples(arena, examples, local_device, group=None):
    """The per-epoch exchange of SURVEY 8(e) end to end on the device: this rank's recorded examples (arena buffers)
    -> RCCL all-gather over xGMI -> appended to an `Examples` set (agz_examples_append_dev), ready for
    Examples.prepare + Trainer.train_dev.  Single process: a device-to-device append.  Every rank ends up with the
    same set in rank order (within a rank: the reference's episode order)."""
    from . import capi
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        examples.append_arena(arena)
        return len(examples)
    local = capi.Examples(examples.ctx, examples.F, examples.H, examples.W, examples.A1)
    local.append_arena(arena)           # canonical (episode) order first, then exchange
    n = len(local)
    dev = torch.device("cuda", local_device)
    p, q, v = local.raw_dev()
    planes = device_tensor(p, (n, examples.F * examples.H * examples.W), dev) if n else torch.zeros((0, examples.F * examples.H * examples.W), device=dev)
    policy = device_tensor(q, (n, examples.A1), dev) if n else torch.zeros((0, examples.A1), device=dev)
    value = device_tensor(v, (n,), dev) if n else torch.zeros((0,), device=dev)
    P, Q, V = all_gather_examples(planes, policy, value, group=group)
    P, Q, V = P.contiguous(), Q.contiguous(), V.contiguous()
    torch.cuda.synchronize(dev)
    examples.append_dev(P.data_ptr(), Q.data_ptr(), V.data_ptr(), int(V.shape[0]))
    examples.ctx.sync()
    local.close()
    return len(examples)


def make_comm(ctx, group=None):
    """An agz_comm (RCCL inside libagz, include/agz.h) for this rank's ctx: rank 0 draws the RCCL unique id, the process
    group ships its 128 bytes (host side, any backend), every rank joins with agz_comm_init_rank.  The exchange itself then
    runs entirely inside libagz: Comm.allgather_examples / Comm.allreduce_trainer — what a Go host calls through cgo."""
    from . import capi
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    box = [capi.Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0, group=group)
    return capi.Comm.init_rank(ctx, world, rank, box[0])


def shard_games(total_games, rank, world):
    """games [lo, hi) owned by `rank` (config #4: 4096 games sharded 512/GPU)"""
    per = total_games // world
    extra = total_games % world
    lo = rank * per + min(rank, extra)
    hi = lo + per + (1 if rank < extra else 0)
    return lo, hi


def allreduce_gradients(trainer, local_device, group=None):
    """Data-parallel dual.Train: ONE all-reduce over the trainer's flat gradient buffer (all learnables of
    the network in one contiguous device buffer), in place.  With RCCL (backend nccl) the device buffer is reduced
    directly over xGMI; with gloo (CPU test rigs) it is staged through host memory.  Returns the world size; follow with
    trainer.apply(lr, grad_scale=1/world) for gradient averaging."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 1
    world = dist.get_world_size(group)
    ptr, n = trainer.grads_dev()
    trainer.ctx.sync()   # the gradients were produced on the ctx stream; the collective runs on torch's
    g = device_tensor(ptr, (n,), torch.device("cuda", local_device))
    if str(dist.get_backend(group)) == "gloo":
        h = g.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        g.copy_(h)
    else:
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=group)
    torch.cuda.synchronize()
    return world
