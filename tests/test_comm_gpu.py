"""GPU: RCCL inside libagz (agz_comm_*).  The GPU box has ONE device, so this is the N = 1 pass-through: the
communicator is a real RCCL communicator (librccl resolved at run time), the example all-gather leaves the rows as they are
and the gradient all-reduce is the identity — also for the per-slice reduction under the backward pass.  The N > 1 exchange logic (counts, rank-ordered union, one collective per
training step) is covered on CPU by tests/test_dist_gloo.py; N > 1 over xGMI has not been run (no multi-GPU box here)."""
import numpy as np
import pytest

import agogo_amd as A
from agogo_amd import capi

pytestmark = pytest.mark.gpu


def test_comm_init_all_single_device_examples_and_gradients_pass_through(ctx):
    comms = A.Comm.init_all([ctx])
    assert len(comms) == 1 and comms[0].rank() == 0 and comms[0].size() == 1
    ex = A.Examples(ctx, 2, 3, 3, 10)
    rng = np.random.default_rng(0)
    p, q, v = rng.normal(size=(7, 18)).astype(np.float32), rng.random((7, 10)).astype(np.float32), rng.choice([-1.0, 0.0, 1.0], 7).astype(np.float32)
    ex.append_host(p, q, v)
    comms[0].allgather_examples(ex)
    gp, gq, gv = ex.get()
    np.testing.assert_array_equal(gp, p)
    np.testing.assert_array_equal(gq, q)
    np.testing.assert_array_equal(gv, v)
    tr = A.Trainer(ctx, 32, 1, 16, 3, 3, 2, 10, 4)
    tr.init_random(3)
    x = rng.choice(np.array([-1, 0.001, 1], np.float32), size=(4, 2, 3, 3)).astype(np.float32)
    pi = np.eye(10, dtype=np.float32)[rng.integers(0, 10, 4)]
    tr.forward_backward(x, pi, np.array([1, -1, 0, 1], np.float32))
    before = [tr.get_grad(i).copy() for i in range(tr.num_params())]
    comms[0].allreduce_trainer(tr)
    ctx.sync()
    for i, b in enumerate(before):
        np.testing.assert_array_equal(tr.get_grad(i), b)
    comms[0].close()


def test_gradient_slices_reduced_under_the_backward_pass_with_the_real_rccl(ctx):
    """agz_trainer_forward_backward_allreduce on a REAL RCCL communicator (n = 1: every all-reduce is the identity, but the 2 + L collectives
    are real ncclAllReduce calls on the communicator's own queue, ordered against the backward pass by events, joined back into the ctx
    stream): gradients and cost equal those of agz_trainer_forward_backward (to the atomics' run-to-run noise, 2e-6 of a tensor's maximum), twice
    in a row (the second step's clears must wait for the first step's reductions), and the averaged step equals the plain one."""
    comm = A.Comm.init_all([ctx])[0]
    K, L, FC, W, H, F, Asp, B = 64, 3, 32, 7, 7, 2, 50, 6
    rng = np.random.default_rng(4)
    t1 = A.Trainer(ctx, K, L, FC, W, H, F, Asp, B)
    t2 = A.Trainer(ctx, K, L, FC, W, H, F, Asp, B)
    t1.init_random(11)
    t2.init_random(11)
    for step in range(2):
        x = rng.choice(np.array([-1, 0.001, 1], np.float32), size=(B, F, H, W)).astype(np.float32)
        pi = np.eye(Asp, dtype=np.float32)[rng.integers(0, Asp, B)]
        v = rng.choice(np.array([-1, 0, 1], np.float32), size=B).astype(np.float32)
        c1 = comm.forward_backward_allreduce(t1, x, pi, v)
        c2 = t2.forward_backward(x, pi, v)
        ctx.sync()
        assert abs(c1 - c2) <= 1e-6 * max(1.0, abs(c2))
        for i in range(t1.num_params()):
            a, b = t1.get_grad(i), t2.get_grad(i)
            name = t1.param_info(i)[0]
            # (two runs of the same step differ in the last bits: the BatchNorm sums and the weight gradient accumulate with atomics)
            assert float(np.abs(a - b).max()) <= 2e-6 * max(float(np.abs(b).max()), 1e-3), name
        t1.apply(0.1, 1.0 / comm.size())
        t2.apply(0.1, 1.0)
    ctx.sync()
    for i in range(t1.num_params()):
        assert float(np.abs(t1.get_param(i) - t2.get_param(i)).max()) <= 2e-6 * max(float(np.abs(t2.get_param(i)).max()), 1e-3)
    t1.close(); t2.close()
    comm.close()


def test_comm_init_rank_with_a_unique_id_and_argument_checks(ctx):
    uid = A.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    c = A.Comm.init_rank(ctx, 1, 0, uid)
    assert c.size() == 1
    c.close()
    with pytest.raises(A.AgzError, match="rank"):
        A.Comm.init_rank(ctx, 2, 5, uid)
    L = capi.lib()
    assert L.agz_comm_init_all(None, 1, None) == -1
    assert L.agz_examples_allgather(None, None) == -1
