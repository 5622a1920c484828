"""CPU, world_size 2, gloo: the N>1 path of the hot path — game sharding (no data-path collective) and the
one exchange step (variable-count all-gather of example records before dual.Train)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from agogo_amd import dist as adist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    r, l, w = adist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    # ranks own disjoint game ranges; examples differ in count per rank
    lo, hi = adist.shard_games(7, rank, world)
    n = (hi - lo) * (rank + 2)
    F, A = 18 * 9, 10
    planes = torch.full((n, F), float(rank + 1))
    policy = torch.full((n, A), 0.1 * (rank + 1))
    value = torch.arange(n, dtype=torch.float32) + 100 * rank
    gp, gpo, gv = adist.all_gather_examples(planes, policy, value)
    q.put((rank, lo, hi, n, gp.shape[0], float(gp.sum()), float(gpo.sum()), gv.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_games_partition():
    for total in (1, 7, 512, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [adist.shard_games(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            for a, b in zip(spans, spans[1:]):
                assert a[1] == b[0]
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert adist.shard_games(4096, 3, 8) == (1536, 2048)  # config #4: 512 games per GPU


def test_all_gather_examples_world2_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, n0, tot0, s0, sp0, v0), (r1, lo1, hi1, n1, tot1, s1, sp1, v1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 4, 4, 7)
    assert n0 == 8 and n1 == 9
    assert tot0 == tot1 == n0 + n1
    F = 18 * 9
    assert s0 == s1 == pytest.approx(n0 * F * 1.0 + n1 * F * 2.0)
    assert v0 == v1 == [float(i) for i in range(n0)] + [100.0 + i for i in range(n1)]


def test_single_process_passthrough():
    p, po, v = torch.zeros(3, 4), torch.zeros(3, 2), torch.zeros(3)
    a, b, c = adist.all_gather_examples(p, po, v)
    assert a is p and b is po and c is v
