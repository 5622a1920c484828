"""CPU, world_size 2, gloo: the host side of the N > 1 path — game sharding (no data-path collective), the communicator-id
exchange in front of agz_comm_init_rank, and bench.py's aggregation over ranks (MAX of the wall time, SUM of the counters).
The exchange itself (agz_examples_allgather / agz_trainer_allreduce) is device code inside libagz: it runs in -m gpu
(tests/test_comm_fake_gpu.py: n = 2 and 3 ranks on one GPU through tests/fake_rccl; tests/test_comm_gpu.py: real RCCL at n = 1)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from agogo_amd import dist as adist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "librccl_fake.so")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), AGZ_RCCL_LIB=FAKE)
    r, l, w = adist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = adist.shard_games(7, rank, world)
    # the 128-byte communicator id: drawn on rank 0 inside libagz (agz_comm_unique_id), shipped over the process group
    uid = adist.exchange_unique_id()
    # bench.py's aggregation: rank r timed 1 + r seconds and counted (100 (r + 1), 7) units
    t_max, sums = adist.reduce_step_timing(1.0 + rank, [100.0 * (rank + 1), 7.0])
    q.put((rank, lo, hi, bytes(uid), t_max, sums))
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


def test_id_exchange_sharding_and_bench_aggregation_world2_gloo():
    assert os.path.exists(FAKE), "tests/fake_rccl/librccl_fake.so is built by `make`"
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
    (r0, lo0, hi0, id0, t0, s0), (r1, lo1, hi1, id1, t1, s1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 4, 4, 7)
    assert len(id0) == 128 and id0 == id1 and any(id0)
    assert t0 == t1 == 2.0                    # MAX over ranks
    assert s0 == s1 == [300.0, 14.0]          # SUM over ranks


def _bench_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    sys.path.insert(0, ROOT)
    import bench
    adist.init_from_env(backend="gloo")
    # rank r: 14.7 + r ms per step over 20 steps, 512 sims per step on rank 0 and 511 on rank 1 (one null simulation)
    dt = (14.7 + rank) * 1e-3 * 20
    agg = bench.aggregate_over_ranks(dt, 512 * 20 - rank * 20, 512 * 20, 512 * 20, rank, world)
    line = bench.headline_fields(agg, world, 20, 4)
    try:
        bench.headline_fields(agg, world + 1, 20, 4)          # a line labelled with another N must not be printed
        refused = False
    except SystemExit:
        refused = True
    q.put((rank, agg, line, refused))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_aggregation_path_world2_gloo():
    """bench.py's own aggregation + headline fields on a world-size-2 process group (gloo): value = all
    ranks' simulations / the slowest rank's time, n_gpus = the ranks that took part, per-rank counters in the line."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, agg, line, refused in res:
        assert refused
        assert agg["per_rank_sims"] == [10240.0, 10220.0] and agg["world_size"] == 2
        assert abs(agg["t_max"] - 15.7e-3 * 20) < 1e-12
        assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["steps"] == 20
        assert abs(line["value"] - 20460.0 / (15.7e-3 * 20)) < 1e-6
        assert abs(line["ms_per_step"] - 15.7) < 1e-9


def test_bench_relaunches_itself_for_n_gpus_without_a_launcher():
    """`python bench.py --gpus N` outside torch.distributed.run must become the N-rank job (it silently measured one
    GPU): the command is the contract's launch line; inside a launcher (RANK / WORLD_SIZE set) and at N = 1 nothing is re-launched."""
    import json
    import subprocess
    import sys
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.launcher_command(["--gpus", "4", "--steps", "5"], 4, {}, port=29511)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "5"]
    assert bench.launcher_command(["--gpus", "4"], 4, {"RANK": "0", "WORLD_SIZE": "4"}) is None
    assert bench.launcher_command([], 1, {}) is None
    # the script's own entry: --print-launch shows what it would exec
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--print-launch"],
                         capture_output=True, text=True, env=env, timeout=120, check=True)
    got = json.loads(out.stdout.strip().splitlines()[-1])
    assert got[cmd.index("--nproc-per-node") + 1] == "2" and got[-4:] == ["--gpus", "2", "--steps", "3"]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--print-launch"], capture_output=True, text=True, env=env, timeout=120, check=True)
    assert json.loads(out.stdout.strip().splitlines()[-1]) is None


def test_single_process_passthrough():
    assert adist.reduce_step_timing(0.5, [3, 4]) == (0.5, [3.0, 4.0])
