"""GPU: the N > 1 branches of libagz's exchange step (/root/reference agogo.go:110-133 runs dual.Train on the union of
all episodes' examples) executed on a ONE-GPU box.  Ranks are processes sharing device 0; librccl is replaced by
tests/fake_rccl/librccl_fake.so (AGZ_RCCL_LIB: the dlopen path libagz honours for any RCCL build), a shared-memory stand-in with
NCCL's semantics for the ten entry points comm.hip binds.  What runs is the PRODUCT code: agz_comm_unique_id / agz_comm_init_rank,
agz_examples_allgather (count exchange, allocation agreement, one grouped set of n broadcasts with rank r the root of its own rows,
received in place, store swap) with uneven counts including a zero-count rank, agz_trainer_allreduce + agz_trainer_apply(1/n), and
agz_trainer_forward_backward_allreduce (the per-slice reduction under the backward pass: bit-identical sums).
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl", "librccl_fake.so")

WORKER = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import agogo_amd as A
rank, n, out, counts = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], [int(c) for c in sys.argv[4].split(",")]
ctx = A.Ctx(0)
idf = out + ".uid"
if rank == 0:
    uid = A.Comm.unique_id()
    with open(idf + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(idf + ".tmp", idf)
else:
    t0 = time.time()
    while not os.path.exists(idf):
        assert time.time() - t0 < 60, "rank 0 never published the unique id"
        time.sleep(0.01)
    uid = open(idf, "rb").read()
comm = A.Comm.init_rank(ctx, n, rank, uid)
assert comm.size() == n and comm.rank() == rank
# --- examples: this rank's rows are recognisable: planes = 1000*rank + row + column/1000
F, H, W, A1 = 2, 3, 3, 10
k = counts[rank]
ex = A.Examples(ctx, F, H, W, A1)
rows = np.arange(k, dtype=np.float32)[:, None]
p = (1000.0 * rank + rows + np.arange(F * H * W, dtype=np.float32)[None, :] / 1000.0).astype(np.float32)
q = (0.5 * rank + rows / 100.0 + np.arange(A1, dtype=np.float32)[None, :]).astype(np.float32)
v = (10.0 * rank + np.arange(k)).astype(np.float32)
if k:
    ex.append_host(p, q, v)
comm.allgather_examples(ex)
gp, gq, gv = ex.get()
# a second gather on the gathered set: every rank now contributes the whole union (n * total rows, rank-major)
comm.allgather_examples(ex)
gp2, gq2, gv2 = ex.get()
# --- data-parallel step: same initial learnables, rank-specific batch, ONE all-reduce over the flat gradient buffer
tr = A.Trainer(ctx, 32, 1, 16, 3, 3, 2, 10, 4)
tr.init_random(3)
rng = np.random.default_rng(100 + rank)
x = rng.choice(np.array([-1, 0.001, 1], np.float32), size=(4, 2, 3, 3)).astype(np.float32)
pi = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 4)]
val = rng.choice(np.array([-1, 0, 1], np.float32), size=4).astype(np.float32)
tr.forward_backward(x, pi, val)
local = [tr.get_grad(i).copy() for i in range(tr.num_params())]
before = [tr.get_param(i).copy() for i in range(tr.num_params())]
comm.allreduce_trainer(tr)
ctx.sync()
summed = [tr.get_grad(i).copy() for i in range(tr.num_params())]
# the same step with the reduction under the backward pass (one collective per slice: heads, then layer L .. 0)
tr2 = A.Trainer(ctx, 32, 1, 16, 3, 3, 2, 10, 4)
tr2.init_random(3)
comm.forward_backward_allreduce(tr2, x, pi, val)
ctx.sync()
summed2 = [tr2.get_grad(i).copy() for i in range(tr2.num_params())]
tr.apply(0.1, 1.0 / n)
ctx.sync()
after = [tr.get_param(i).copy() for i in range(tr.num_params())]
np.savez(out + ".r%d.npz" % rank, gp=gp, gq=gq, gv=gv, gp2=gp2, gv2=gv2, own_p=p, own_q=q, own_v=v,
         **{"local%d" % i: g for i, g in enumerate(local)}, **{"sum%d" % i: g for i, g in enumerate(summed)}, **{"osum%d" % i: g for i, g in enumerate(summed2)},
         **{"before%d" % i: g for i, g in enumerate(before)}, **{"after%d" % i: g for i, g in enumerate(after)}, nparams=tr.num_params())
comm.close()
"""


@pytest.mark.parametrize("counts", [(4, 7), (5, 0, 3), (0, 6), (2, 1, 0)])
def test_allgather_and_allreduce_over_n_ranks_on_one_gpu(counts, tmp_path):
    assert os.path.exists(FAKE), "tests/fake_rccl/librccl_fake.so is built by `make` (__graft_entry__.build)"
    n = len(counts)
    out = str(tmp_path / "x")
    env = dict(os.environ, AGZ_RCCL_LIB=FAKE)
    procs = [subprocess.Popen([sys.executable, "-c", WORKER, str(r), str(n), out, ",".join(map(str, counts))], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(n)]
    logs = []
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-3000:])
    R = [np.load(out + ".r%d.npz" % r) for r in range(n)]
    # the union, rank after rank, on every rank
    exp_p = np.concatenate([R[r]["own_p"] for r in range(n)])
    exp_q = np.concatenate([R[r]["own_q"] for r in range(n)])
    exp_v = np.concatenate([R[r]["own_v"] for r in range(n)])
    assert exp_v.size == sum(counts)
    for r in range(n):
        np.testing.assert_array_equal(R[r]["gp"], exp_p)
        np.testing.assert_array_equal(R[r]["gq"], exp_q)
        np.testing.assert_array_equal(R[r]["gv"], exp_v)
        np.testing.assert_array_equal(R[r]["gp2"], np.concatenate([exp_p] * n))
        np.testing.assert_array_equal(R[r]["gv2"], np.concatenate([exp_v] * n))
    # gradients: the sum of what each rank computed alone (rank order, fp32), identical on every rank; averaged SGD step
    npar = int(R[0]["nparams"])
    for i in range(npar):
        s = R[0]["local%d" % i].astype(np.float32).copy()
        for r in range(1, n):
            s = (s + R[r]["local%d" % i]).astype(np.float32)
        assert any(np.abs(R[r]["local%d" % i]).max() > 0 for r in range(n)) or s.size == 0
        for r in range(n):
            np.testing.assert_array_equal(R[r]["sum%d" % i], s)
            # agz_trainer_forward_backward_allreduce (one collective per slice, under the backward pass): the same bits
            np.testing.assert_array_equal(R[r]["osum%d" % i], s)
            np.testing.assert_array_equal(R[r]["before%d" % i], R[0]["before%d" % i])
            np.testing.assert_array_equal(R[r]["after%d" % i], R[0]["after%d" % i])
        np.testing.assert_allclose(R[0]["after%d" % i], R[0]["before%d" % i] - np.float32(0.1 / n) * s, rtol=2e-6, atol=1e-7)
    # rank-specific batches really differed
    assert not np.array_equal(R[0]["local0"], R[1]["local0"])


FAIL_WORKER = r"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import agogo_amd as A
rank, n, out, bad_rank, bad_slice = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5])
ctx = A.Ctx(0)
idf = out + ".uid"
if rank == 0:
    uid = A.Comm.unique_id()
    with open(idf + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(idf + ".tmp", idf)
else:
    t0 = time.time()
    while not os.path.exists(idf):
        assert time.time() - t0 < 60
        time.sleep(0.01)
    uid = open(idf, "rb").read()
comm = A.Comm.init_rank(ctx, n, rank, uid)
tr = A.Trainer(ctx, 32, 2, 16, 3, 3, 2, 10, 4)          # slices: heads, layer 2, layer 1, layer 0
tr.init_random(3)
rng = np.random.default_rng(100 + rank)
x = rng.choice(np.array([-1, 0.001, 1], np.float32), size=(4, 2, 3, 3)).astype(np.float32)
pi = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 4)]
val = rng.choice(np.array([-1, 0, 1], np.float32), size=4).astype(np.float32)
tr.forward_backward(x, pi, val)
local = [tr.get_grad(i).copy() for i in range(tr.num_params())]
if rank == bad_rank:
    comm.debug_fail_slice(bad_slice)
msg = ""
try:
    comm.forward_backward_allreduce(tr, x, pi, val)
except A.AgzError as e:
    msg = str(e)
ctx.sync()
# the next step is an ordinary one on every rank
comm.forward_backward_allreduce(tr, x, pi, val)
ctx.sync()
summed = [tr.get_grad(i).copy() for i in range(tr.num_params())]
np.savez(out + ".r%d.npz" % rank, msg=np.array(msg), nparams=tr.num_params(),
         **{"local%d" % i: g for i, g in enumerate(local)}, **{"sum%d" % i: g for i, g in enumerate(summed)})
comm.close()
"""


@pytest.mark.parametrize("n,bad_rank,bad_slice", [(2, 1, 1), (3, 0, 0), (2, 0, 3)])
def test_a_rank_that_fails_inside_the_data_parallel_step_does_not_hang_its_peers(n, bad_rank, bad_slice, tmp_path):
    """agz_trainer_forward_backward_allreduce issues L + 2 collectives from inside the backward pass; a rank that
    failed after the first one used to return with the rest un-entered — its peers blocked for ever.  Now the failing rank still enters every
    collective of the step and all ranks exchange a status word: the call fails on EVERY rank (AGZ_E_PEER = -7 on the healthy ones), nobody
    hangs, and the following step reduces normally (bit-identical to the sum of the ranks' own gradients).  Failure injected by
    agz_comm_debug_fail_slice before the first, a middle and the last slice."""
    assert os.path.exists(FAKE)
    out = str(tmp_path / "f")
    env = dict(os.environ, AGZ_RCCL_LIB=FAKE)
    procs = [subprocess.Popen([sys.executable, "-c", FAIL_WORKER, str(r), str(n), out, str(bad_rank), str(bad_slice)], cwd=ROOT, env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(n)]
    logs = []
    for pr in procs:
        try:
            o, _ = pr.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode(errors="replace"))
    for r, pr in enumerate(procs):
        assert pr.returncode == 0, "rank %d failed:\n%s" % (r, logs[r][-3000:])
    R = [np.load(out + ".r%d.npz" % r) for r in range(n)]
    for r in range(n):
        msg = str(R[r]["msg"])
        if r == bad_rank:
            assert "(-4)" in msg and "injected failure before slice %d" % bad_slice in msg, msg
        else:
            assert "(-7)" in msg and "another rank failed" in msg, msg
    for i in range(int(R[0]["nparams"])):
        s_ = R[0]["local%d" % i].astype(np.float32).copy()
        for r in range(1, n):
            s_ = (s_ + R[r]["local%d" % i]).astype(np.float32)
        for r in range(n):
            np.testing.assert_array_equal(R[r]["sum%d" % i], s_)


def test_fake_rccl_is_test_infrastructure_only():
    """the product never names the double: it only honours AGZ_RCCL_LIB"""
    for d, _, fs in os.walk(os.path.join(ROOT, "agogo_amd")):
        for f in fs:
            if f.endswith((".hip", ".hpp", ".py", ".h")):
                assert "fake_rccl" not in open(os.path.join(d, f), errors="replace").read().replace("tests/fake_rccl's", ""), f


def test_bench_two_ranks_end_to_end_through_the_self_relaunch(tmp_path):
    """`python bench.py --gpus 2` OUTSIDE a launcher — the path the first multi-GPU run takes if the driver calls the
    script plainly: it re-executes itself under torch.distributed.run (two ranks, 127.0.0.1 rendezvous), both ranks run the timed
    region on GPU 0 (--shared-gpu: gloo for the process group), the example exchange runs inside libagz over the in-tree RCCL double
    (AGZ_RCCL_LIB), and rank 0 prints ONE line whose self-checks hold: n_gpus = 2, both ranks simulated, rccl_ranks = 2 and
    rows_gathered = the sum of the ranks' own rows.  The line is kept under gpurun_out/ (profiles/r05/bench_n2_two_ranks_one_gpu.json)."""
    import json
    env = dict(os.environ, AGZ_RCCL_LIB=FAKE, MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--shared-gpu", "--games", "32", "--budget", "16", "--steps", "2",
           "--warmup", "1", "--L", "4", "--cpu-threads", "8", "--cpu-baseline-seconds", "2", "--no-games-leg", "--no-go9-leg", "--no-latency-leg",
           "--no-train-leg", "--no-f32-leg"]
    pr = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in pr.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["scaling"] == "weak"
    assert len(d["per_rank_sims"]) == 2 and min(d["per_rank_sims"]) > 0
    assert abs(sum(d["per_rank_sims"]) - d["value"] * d["ms_per_step"] * d["steps"] * 1e-3) < 1.0
    assert d["rccl_ranks"] == 2
    assert d["rows_check"] == "ok" and len(d["rows_per_rank"]) == 2 and sum(d["rows_per_rank"]) == d["rows_gathered"] > 0
    assert min(d["rows_per_rank"]) > 0                         # both ranks' arenas recorded examples (a move boundary inside the run)
    # the N > 1 line is COMPLETE — the CPU baseline (rank 0, once; the other rank waits in the closing barrier), the
    # roofline object, and every rank's device memory in use
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] and cb["value"] > 0 and cb["cores"] == 8 and "rank 0 only" in cb["ran_on"]
    assert d["roofline"]["bound"] in ("hbm", "mfma") and d["roofline"]["frac"] > 0 and d["roofline"]["end_to_end"]["frac_of_fp32_mfma_peak"] > 0
    assert len(d["config"]["hbm_used_bytes_per_rank"]) == 2 and min(d["config"]["hbm_used_bytes_per_rank"]) > 0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_n2_two_ranks_one_gpu.json"), "w") as f:
        json.dump(d, f, indent=1)


def test_bench_eight_ranks_rehearsal_on_one_gpu(tmp_path):
    """the driver's N = 8 launch line rehearsed before hardware sees it — `python -m torch.distributed.run --nproc-per-node 8
    ... bench.py --gpus 8` with all eight ranks on GPU 0 (--shared-gpu: gloo process group, the in-tree RCCL double for libagz's exchange):
    ports, environment, the eight-way aggregation (n_gpus = 8, every rank simulated, rows_gathered = the sum of eight ranks' rows, eight
    communicator ranks), per-rank device memory, and a clean exit of all eight processes."""
    import json
    import socket
    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    env = dict(os.environ, AGZ_RCCL_LIB=FAKE)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--shared-gpu", "--games", "16", "--budget", "8", "--steps", "2", "--warmup", "1", "--L", "2",
           "--no-cpu-baseline", "--no-games-leg", "--no-go9-leg", "--no-latency-leg", "--no-train-leg", "--no-f32-leg"]
    pr = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert pr.returncode == 0, pr.stderr.decode(errors="replace")[-3000:]
    lines = [l for l in pr.stdout.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, pr.stdout.decode()[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["world_size"] == 8 and d["scaling"] == "weak"
    assert len(d["per_rank_sims"]) == 8 and min(d["per_rank_sims"]) > 0
    assert abs(sum(d["per_rank_sims"]) - d["value"] * d["ms_per_step"] * d["steps"] * 1e-3) < 1.0
    assert d["rccl_ranks"] == 8 and d["rows_check"] == "ok" and len(d["rows_per_rank"]) == 8 and sum(d["rows_per_rank"]) == d["rows_gathered"] > 0
    assert len(d["config"]["hbm_used_bytes_per_rank"]) == 8 and min(d["config"]["hbm_used_bytes_per_rank"]) > 0
    assert d["roofline"]["frac"] > 0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_n8_eight_ranks_one_gpu.json"), "w") as f:
        json.dump(d, f, indent=1)


def test_comm_init_all_argument_paths_for_more_than_one_rank(ctx):
    """agz_comm_init_all's n > 1 argument handling without a second GPU — a NULL entry in the ctx list, the same device
    twice (one rank per GPU), n out of range, NULL output: every one a clean AGZ_E_INVALID with a message, nothing allocated, no call into
    RCCL (the checks come first), and the n = 1 path still works afterwards."""
    import ctypes as C
    import agogo_amd as A
    from agogo_amd import capi
    L = capi.lib()
    E_INVALID = -1                                             # AGZ_E_INVALID (include/agz.h)
    last = lambda: L.agz_last_error().decode(errors="replace")
    ctx2 = A.Ctx(0)                                            # a second context on the SAME device
    two = (C.c_void_p * 2)(ctx.h, ctx2.h)
    out = (C.c_void_p * 2)()
    assert L.agz_comm_init_all(two, 2, out) == E_INVALID
    assert "appears twice" in last()
    assert not out[0] and not out[1]
    hole = (C.c_void_p * 2)(ctx.h, None)
    assert L.agz_comm_init_all(hole, 2, out) == E_INVALID
    assert "ctxs[1] is NULL" in last()
    assert L.agz_comm_init_all(two, 0, out) == E_INVALID
    assert L.agz_comm_init_all(two, 65, out) == E_INVALID
    assert L.agz_comm_init_all(two, 2, None) == E_INVALID
    comms = A.Comm.init_all([ctx])
    assert comms[0].size() == 1
    comms[0].close()
    ctx2.close()
