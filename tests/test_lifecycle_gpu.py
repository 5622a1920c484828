"""GPU: handle lifecycle — repeated create/use/destroy of every handle type returns device memory, errors are reported
through agz_last_error, and misuse fails loudly instead of crashing."""
import numpy as np
import pytest
import torch

import agogo_amd as A
from agogo_amd import capi

pytestmark = pytest.mark.gpu


def _cycle(ctx):
    net = A.Net(ctx, 64, 2, 64, 9, 9, 18, 82, bn_mode=capi.BN_IDENTITY)
    net.init_random(1)
    net.commit()
    net.set_compute_mode(capi.COMPUTE_BF16X3 | capi.COMPUTE_FORCE)
    x = np.zeros((40, 18, 9, 9), np.float32)
    net.infer(x)          # throughput regime (bf16x3 weights resident)
    net.infer(x[:1])      # latency regime (split-K workspace + spread heads scratch)
    arena = A.Arena(ctx, capi.GAME_WQ, 9, 9, 0, 7.5, encoder=capi.ENC_WQ, n_games=8, Budget=6)
    arena.set_inferencer(0, capi.INF_NET, net)
    arena.set_inferencer(1, capi.INF_NET, net)
    arena.reset()
    arena.play(3, True)
    ex = A.Examples(ctx, 18, 9, 9, 82)
    ex.append_arena(arena)
    ex.augment_rotate()
    ex.prepare(8, 0, seed=1)
    tr = A.Trainer(ctx, 32, 1, 16, 9, 9, 18, 82, 8)
    tr.init_random(2)
    xd, pd, vd, rows, b = ex.tensors_dev()
    if b:
        tr.train_dev(xd, pd, vd, b, 1, seed=3)
    for h in (tr, ex, arena, net):
        h.close()


def test_create_destroy_cycles_do_not_leak_device_memory(ctx):
    _cycle(ctx)  # warm: allocator pools, code objects
    ctx.sync()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(6):
        _cycle(ctx)
    ctx.sync()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 32 * 1024 * 1024, "leaked %.1f MB over 6 cycles" % ((free0 - free1) / 2**20)


def test_misuse_fails_loudly(ctx):
    with pytest.raises(A.AgzError):
        A.Net(ctx, 0, 1, 8, 3, 3, 2, 10)                      # dual.Config.IsValid
    net = A.Net(ctx, 32, 1, 8, 3, 3, 2, 10)
    with pytest.raises(A.AgzError, match="commit"):
        net.infer(np.zeros((1, 2, 3, 3), np.float32))          # infer before commit
    with pytest.raises(A.AgzError):
        net.set_compute_mode(7)
    with pytest.raises(A.AgzError):
        A.Arena(ctx, capi.GAME_MNK, 3, 3, 3, n_games=2, PUCT=1.5)   # mcts.Config.IsValid: 0 < PUCT <= 1
    arena = A.Arena(ctx, capi.GAME_MNK, 3, 3, 3, n_games=2, Budget=4)
    with pytest.raises(A.AgzError):
        arena.set_inferencer(0, capi.INF_NET, None)            # NET inferencer without a net
    arena.close()
    net.close()
