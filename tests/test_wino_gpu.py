"""GPU: the Winograd F(4x4,3x3) path (AGZ_COMPUTE_WINO, agogo_amd/csrc/conv_wino.hpp).  Stage by stage against numpy (input
transform Bt d B, the 36 transform-domain GEMMs), then whole networks against the oracle and the fp32-MFMA path with the
same tolerance every other arithmetic mode meets."""
import numpy as np
import pytest

import agogo_amd as A
from agogo_amd import capi
from test_net_gpu import make_pair, rand_planes, POL_ATOL, POL_RTOL, VAL_ATOL

pytestmark = pytest.mark.gpu

BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
               [0, 4, 0, -5, 0, 1]], np.float64)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
              [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)


def numpy_stages(x, w):
    """x [B,H,W,C], w [N,C,3,3] (float64 arithmetic) -> V [36,T,C], M [36,T,N], y [B,H,W,N] (At M A, cropped)"""
    B, H, W, C = x.shape
    N = w.shape[0]
    nty, ntx = (H + 3) // 4, (W + 3) // 4
    xp = np.zeros((B, 4 * nty + 2, 4 * ntx + 2, C))
    xp[:, 1:H + 1, 1:W + 1] = x
    U = np.einsum("ai,ncij,bj->abnc", G, w.astype(np.float64), G).astype(np.float32).astype(np.float64)   # rounded once, like commit
    V = np.zeros((36, B * nty * ntx, C))
    t = 0
    for b in range(B):
        for ty in range(nty):
            for tx in range(ntx):
                d = xp[b, 4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6]
                V[:, t] = np.einsum("ai,ijc,bj->abc", BT, d, BT).reshape(36, C)
                t += 1
    M = np.einsum("ptc,pnc->ptn", V, U.reshape(36, N, C))
    Y = np.einsum("ka,abtn,lb->tkln", AT, M.reshape(6, 6, -1, N), AT)
    y = np.zeros((B, 4 * nty, 4 * ntx, N))
    t = 0
    for b in range(B):
        for ty in range(nty):
            for tx in range(ntx):
                y[b, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4] = Y[t]
                t += 1
    return V, M, y[:, :H, :W]


def direct_conv(x, w):
    B, H, W, C = x.shape
    xp = np.zeros((B, H + 2, W + 2, C))
    xp[:, 1:-1, 1:-1] = x
    y = np.zeros((B, H, W, w.shape[0]))
    for ky in range(3):
        for kx in range(3):
            y += np.einsum("bhwc,nc->bhwn", xp[:, ky:ky + H, kx:kx + W], w[:, :, ky, kx].astype(np.float64))
    return y


@pytest.mark.parametrize("B,H,W,C,N", [(2, 7, 5, 16, 40), (3, 19, 19, 64, 128), (1, 4, 4, 32, 130), (5, 9, 9, 48, 256)])
def test_stages_match_numpy(ctx, B, H, W, C, N):
    rng = np.random.default_rng(B * 1000 + C)
    x = np.maximum(rng.normal(0, 1, (B, H, W, C)), 0).astype(np.float32)
    w = (rng.uniform(-1, 1, (N, C, 3, 3)) * np.sqrt(6 / (9 * C + 9 * N))).astype(np.float32)
    V, M = A.wino_stages(ctx, x, w)
    Vn, Mn, yn = numpy_stages(x.astype(np.float64), w)
    # the numpy restatement itself: At M A equals the direct convolution
    np.testing.assert_allclose(yn, direct_conv(x.astype(np.float64), w), atol=1e-4 * np.abs(yn).max())
    np.testing.assert_allclose(V, Vn, atol=2e-6 * np.abs(Vn).max(), rtol=0)          # small-integer transform in fp32
    np.testing.assert_allclose(M, Mn, atol=4e-6 * np.abs(Mn).max(), rtol=0)          # fp32-grade products, fp32 accumulation over C


@pytest.mark.parametrize("K,L,FC,W,H,F,Aspace,B,bn_mode", [
    (64, 2, 128, 9, 9, 18, 82, 70, 0),      # partial last row tile
    (128, 2, 64, 9, 9, 18, 82, 33, 2),
    (256, 2, 128, 19, 19, 18, 362, 12, 2),  # BASELINE width and board: 5x5 tiles, the last tile row/column hangs over the edge
    (64, 5, 64, 9, 9, 18, 82, 64, 1),       # deeper tower, running-stats BN
    (192, 1, 64, 7, 6, 2, 8, 37, 2),        # K=192, 6x7 board (2x2 tiles, both ragged)
    (128, 1, 32, 5, 5, 2, 26, 90, 2),       # 5x5 board
])
@pytest.mark.parametrize("wmode", [A.capi.COMPUTE_WINO, A.capi.COMPUTE_WINO_H2])
def test_wino_networks_match_oracle_and_f32(ctx, K, L, FC, W, H, F, Aspace, B, bn_mode, wmode):
    onet, gnet = make_pair(ctx, K, L, FC, W, H, F, Aspace, bn_mode)
    x = rand_planes(B, F, H, W, seed=K + B)
    pol_f, val_f = gnet.infer(x)
    gnet.set_compute_mode(wmode | A.capi.COMPUTE_FORCE)
    pol_g, val_g = gnet.infer(x)
    assert not np.array_equal(pol_g, pol_f)          # really a different arithmetic path
    idx = [0, B // 2, B - 1]
    pol_o, val_o = onet.infer(x[idx])
    np.testing.assert_allclose(pol_g[idx], pol_o, atol=POL_ATOL, rtol=POL_RTOL)
    np.testing.assert_allclose(val_g[idx], val_o, atol=VAL_ATOL)
    np.testing.assert_allclose(pol_g, pol_f, atol=POL_ATOL, rtol=POL_RTOL)
    np.testing.assert_allclose(val_g, val_f, atol=VAL_ATOL)
    print("wino[%d] max |dpol| vs f32: %.3e  vs oracle: %.3e   f32 vs oracle: %.3e" % (
        wmode, np.abs(pol_g - pol_f).max(), np.abs(pol_g[idx] - pol_o).max(), np.abs(pol_f[idx] - pol_o).max()))


@pytest.mark.parametrize("scale", [1e-6, 1e-3, 1.0, 3e3, 1e6])
def test_wino_h2_range_management_and_batch_independence(ctx, scale):
    """fp16x2 in the transform domain: the per-board power-of-two scale comes from a bound (|Bt d B| <= 100 max|d|), so no input
    magnitude can overflow fp16 — inputs scaled from 1e-6 to 1e6 (the degenerate BatchNorm multiplies by 316 per layer on top)
    stay finite and inside the tolerance; and a board's result does not depend on its batch neighbours (one loud and one
    all-zero board next to it), bit for bit."""
    onet, gnet = make_pair(ctx, 64, 3, 32, 9, 9, 18, 82, 0)
    x = (rand_planes(40, 18, 9, 9, seed=3) * scale).astype(np.float32)
    pol_f, val_f = gnet.infer(x)
    gnet.set_compute_mode(A.capi.COMPUTE_WINO_H2 | A.capi.COMPUTE_FORCE)
    pol_g, val_g = gnet.infer(x)
    assert np.all(np.isfinite(pol_g)) and np.all(np.isfinite(val_g))
    np.testing.assert_allclose(pol_g, pol_f, atol=POL_ATOL, rtol=POL_RTOL)
    np.testing.assert_allclose(val_g, val_f, atol=VAL_ATOL)
    y = x.copy()
    y[1] *= 1000.0
    y[2] = 0.0
    pol_h, val_h = gnet.infer(y)
    np.testing.assert_array_equal(pol_h[0], pol_g[0])
    np.testing.assert_array_equal(pol_h[3:], pol_g[3:])
    np.testing.assert_array_equal(val_h[3:], val_g[3:])
    assert np.all(np.isfinite(pol_h))


def test_compute_auto_takes_the_measured_mode(ctx):
    """AGZ_COMPUTE_AUTO: the Winograd fp16x2 tower where its weights exist (K a multiple of 64), else the bf16x3 split kernels —
    each branch bit-identical to the explicitly chosen mode and inside the oracle tolerance."""
    for (K, S, B, explicit) in ((128, 9, 300, A.capi.COMPUTE_WINO_H2), (32, 9, 600, A.capi.COMPUTE_BF16X3)):
        onet, gnet = make_pair(ctx, K, 2, 32, S, S, 18, S * S + 1, 2)
        x = rand_planes(B, 18, S, S, seed=K)
        gnet.set_compute_mode(explicit)
        pe, ve = gnet.infer(x)
        gnet.set_compute_mode(A.capi.COMPUTE_F32_MFMA)
        pf, vf = gnet.infer(x)
        gnet.set_compute_mode(A.capi.COMPUTE_AUTO)
        pa, va = gnet.infer(x)
        np.testing.assert_array_equal(pa, pe)
        np.testing.assert_array_equal(va, ve)
        if K == 128:
            assert not np.array_equal(pa, pf)        # the batch fills the chip: really the Winograd arithmetic
        idx = [0, B // 2, B - 1]
        po, vo = onet.infer(x[idx])
        np.testing.assert_allclose(pa[idx], po, atol=POL_ATOL, rtol=POL_RTOL)
        np.testing.assert_allclose(va[idx], vo, atol=VAL_ATOL)
        gnet.commit()                                 # a re-commit under AUTO rebuilds what AUTO needs
        pc, vc = gnet.infer(x)
        np.testing.assert_array_equal(pc, pa)


def test_wino_recommit_and_mode_round_trip(ctx):
    """weights are rebuilt on commit; switching modes back and forth keeps every mode's own result"""
    onet, gnet = make_pair(ctx, 64, 1, 32, 9, 9, 18, 82, 2)
    x = rand_planes(24, 18, 9, 9, seed=4)
    gnet.set_compute_mode(A.capi.COMPUTE_WINO | A.capi.COMPUTE_FORCE)
    p1, v1 = gnet.infer(x)
    gnet.set_compute_mode(A.capi.COMPUTE_F32_MFMA)
    pf, vf = gnet.infer(x)
    gnet.set_compute_mode(A.capi.COMPUTE_WINO | A.capi.COMPUTE_FORCE)
    p2, v2 = gnet.infer(x)
    np.testing.assert_array_equal(p1, p2)
    np.testing.assert_array_equal(v1, v2)
    w = gnet.get_param(3)
    gnet.set_param(3, (w * 0.5).astype(np.float32))     # first dual block, branch a filter
    gnet.commit()
    p3, _ = gnet.infer(x)
    gnet.set_compute_mode(A.capi.COMPUTE_F32_MFMA)
    pf3, _ = gnet.infer(x)
    assert not np.array_equal(p3, p1)
    np.testing.assert_allclose(p3, pf3, atol=POL_ATOL, rtol=POL_RTOL)


def test_wino_board_chunks_in_a_subprocess():
    """AGZ_WINO_CHUNK (read once per process) forces the chunked schedule that otherwise only starts above ~4660 boards:
    same results as one launch over the whole batch, bitwise (tiles are independent GEMM rows)."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import agogo_amd as A
from test_net_gpu import make_pair, rand_planes
ctx = A.Ctx(0)
onet, gnet = make_pair(ctx, 64, 2, 32, 9, 9, 18, 82, 2)
gnet.set_compute_mode(A.capi.COMPUTE_WINO | A.capi.COMPUTE_FORCE)
pol, val = gnet.infer(rand_planes(37, 18, 9, 9, seed=11))
np.save(sys.argv[1], np.concatenate([pol.ravel(), val.ravel()]))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for chunk in ("0", "8"):
        path = os.path.join(root, "gpurun_out", "wino_chunk_%s.npy" % chunk)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        env = dict(os.environ, AGZ_WINO_CHUNK=chunk)
        subprocess.run([sys.executable, "-c", code, path], check=True, cwd=root, env=env, timeout=120)
        outs.append(np.load(path))
    np.testing.assert_array_equal(outs[0], outs[1])


def test_wino_h2_two_queue_tower_is_bit_identical(ctx):
    """agz_net_set_tower_queues: the batch split into two half chains on two HIP streams (the default from 256 boards on) gives the
    same bits as one chain over the whole batch — boards are independent (per-board ranges) — at a batch the auto rule splits (256)
    and at an odd one, forced."""
    onet, gnet = make_pair(ctx, 64, 2, 32, 9, 9, 18, 82, 2)
    gnet.set_compute_mode(A.capi.COMPUTE_WINO_H2 | A.capi.COMPUTE_FORCE)
    for B in (256, 77):
        x = rand_planes(B, 18, 9, 9, seed=B)
        gnet.set_tower_queues(1)
        p1, v1 = gnet.infer(x)
        gnet.set_tower_queues(2)
        p2, v2 = gnet.infer(x)
        gnet.set_tower_queues(0)
        p0, v0 = gnet.infer(x)
        np.testing.assert_array_equal(p1, p2)
        np.testing.assert_array_equal(v1, v2)
        np.testing.assert_array_equal(p1, p0)
        np.testing.assert_array_equal(v1, v0)
    with pytest.raises(A.AgzError):
        gnet.set_tower_queues(3)


@pytest.mark.parametrize("K,S,B", [(128, 9, 37), (384, 9, 5), (512, 9, 4), (256, 19, 33), (128, 7, 21)])
def test_wino_h2_chained_block_forms_and_k_extents(ctx, K, S, B):
    """The chained block (conv_wino_h2c.hpp: GEMM by LDS DMA + fused, pipelined out->in kernel; range words from the commit-time bound)
    on every K extent it is instantiated for (C/32 = 4, 8, 12, 16), F(5x5,3x3) and F(4x4,3x3) boards, ragged 16-row groups: against
    the oracle with the tolerance of every network test; its plain (non-pipelined) out->in kernel and the three-kernel block agree
    with it to rounding; one queue, two queues and a repeated call give the same bits."""
    L, F = 3, 18
    onet, gnet = make_pair(ctx, K, L, 32, S, S, F, S * S + 1, 2, seed=K + S)
    assert capi.wino_h2_chained(S, S, K) == 1
    x = rand_planes(B, F, S, S, seed=B)
    gnet.set_compute_mode(A.capi.COMPUTE_WINO_H2 | A.capi.COMPUTE_FORCE)
    outs = {}
    for form in (-1, 1, 0):                       # chained (default), chained with the plain out->in kernel, three-kernel block
        gnet.set_wino_h2_form(form)
        outs[form] = gnet.infer(x)
    gnet.set_wino_h2_form(-1)
    p2, v2 = gnet.infer(x)
    np.testing.assert_array_equal(p2, outs[-1][0])
    np.testing.assert_array_equal(v2, outs[-1][1])
    if B >= 8:
        gnet.set_tower_queues(1)
        p1q, v1q = gnet.infer(np.concatenate([x] * 8)[:max(64, B)])      # >= 64 boards: two queues available
        gnet.set_tower_queues(2)
        p2q, v2q = gnet.infer(np.concatenate([x] * 8)[:max(64, B)])
        gnet.set_tower_queues(0)
        np.testing.assert_array_equal(p1q, p2q)
        np.testing.assert_array_equal(v1q, v2q)
        np.testing.assert_array_equal(p1q[:B], outs[-1][0])                # and a board's result does not depend on the batch around it
    nb = min(B, 4)
    po, vo = onet.infer(x[:nb])
    for form, (p, v) in outs.items():
        assert np.all(np.isfinite(p)) and np.all(np.isfinite(v))
        np.testing.assert_allclose(p[:nb], po, atol=POL_ATOL, rtol=POL_RTOL, err_msg="form %d" % form)
        np.testing.assert_allclose(v[:nb], vo, atol=VAL_ATOL, err_msg="form %d" % form)
    assert not np.array_equal(outs[-1][0], outs[0][0])                     # the chained form really is another schedule of the arithmetic
    np.testing.assert_allclose(outs[-1][0], outs[0][0], atol=POL_ATOL, rtol=POL_RTOL)
    print("\n[chained block] K=%d %dx%d B=%d: max|dpol| vs oracle chained %.2e plain %.2e three-kernel %.2e"
          % (K, S, S, B, np.abs(outs[-1][0][:nb] - po).max(), np.abs(outs[1][0][:nb] - po).max(), np.abs(outs[0][0][:nb] - po).max()))


H2_KNOBS = [
    {"AGZ_WINO_H2_TM": "4"},                                       # F(4x4,3x3) on boards where F(5x5,3x3) is the default
    {"AGZ_WINO_H2_TM": "5"},                                       # F(5x5,3x3) on boards where F(4x4,3x3) is the default (9x9)
    {"AGZ_WINO_H2_CHUNK": "16"},                                   # board chunks
    {"AGZ_WINO_H2_CHUNK": "16", "AGZ_WINO_H2_QUEUES": "2"},        # board chunks on two queues
    {"AGZ_WINO_H2_QUEUES": "1"},                                   # one queue whatever the batch
    {"AGZ_WINO_H2_FORM": "0"},                                     # the three-kernel block instead of the chained one
    {"AGZ_WINO_H2_GEMM": "2"},                                     # the persistent GEMM (weight slab stationary in registers; K = 256)
    {"AGZ_WINO_H2_GEMM": "2", "AGZ_WINO_H2_CHUNK": "16", "AGZ_WINO_H2_QUEUES": "2"},   # ... on short per-team unit lists, two queues
    {"AGZ_WINO_H2_GEMM": "65"},                                    # the default GEMM with round 4's store policy (default: non-temporal M / V2c stores)
]


@pytest.mark.parametrize("knobs", H2_KNOBS, ids=lambda k: ",".join("%s=%s" % (a[12:], b) for a, b in k.items()))
def test_wino_h2_tuning_knobs_in_a_subprocess(knobs):
    """The AGZ_WINO_H2_* environment switches (include/agz.h, read once per process) select the other tile size and other
    schedules of the same arithmetic: every one of them stays inside the network tolerance against the fp32-MFMA path, and the
    ones that only move data (chunks, queues) reproduce the default bit for bit."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import agogo_amd as A
from test_net_gpu import make_pair, rand_planes
ctx = A.Ctx(0)
out = []
for (K, L, S, B) in ((256, 2, 19, 70), (64, 2, 9, 37)):
    onet, gnet = make_pair(ctx, K, L, 32, S, S, 18, S * S + 1, 2)
    x = rand_planes(B, 18, S, S, seed=11)
    pf, vf = gnet.infer(x)
    gnet.set_compute_mode(A.capi.COMPUTE_WINO_H2 | A.capi.COMPUTE_FORCE)
    pol, val = gnet.infer(x)
    out += [pf.ravel(), vf.ravel(), pol.ravel(), val.ravel()]
np.save(sys.argv[1], np.concatenate(out))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    base = {k: v for k, v in os.environ.items() if not k.startswith("AGZ_WINO_H2_")}
    for tag, env in (("default", base), ("knob", dict(base, **knobs))):
        path = os.path.join(root, "gpurun_out", "wino_h2_%s.npy" % tag)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        subprocess.run([sys.executable, "-c", code, path], check=True, cwd=root, env=env, timeout=180)
        outs.append(np.load(path))
    n = 0
    for (S, B) in ((19, 70), (9, 37)):
        A_ = S * S + 1
        pf = outs[1][n:n + B * A_]; vf = outs[1][n + B * A_:n + B * A_ + B]
        pg = outs[1][n + B * A_ + B:n + 2 * B * A_ + B]; vg = outs[1][n + 2 * B * A_ + B:n + 2 * B * A_ + 2 * B]
        np.testing.assert_allclose(pg, pf, atol=POL_ATOL, rtol=POL_RTOL)
        np.testing.assert_allclose(vg, vf, atol=VAL_ATOL)
        n += 2 * B * A_ + 2 * B
    if "AGZ_WINO_H2_TM" not in knobs and "AGZ_WINO_H2_FORM" not in knobs:      # same arithmetic in the same order: only the data movement differs
        np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.gpu
@pytest.mark.parametrize("S,L,B", [(19, 3, 70), (19, 2, 1), (19, 2, 3), (19, 2, 16), (19, 3, 300), (9, 3, 37), (9, 2, 512)])
def test_persistent_gemm_is_bit_identical(ctx, S, L, B):
    """wino_gemm_h2p_kernel (conv_wino_h2c.hpp; agz_net_set_wino_h2_gemm(net, 2)): per accumulator the same MFMA sequence as
    wino_gemm_h2g_kernel, so policy and value must be BIT-IDENTICAL — on short team lists (B = 1, 3, 16: some teams get no units), ragged
    last m-tiles (B = 70, 300), 9x9 boards (four boards per 16-tile group), and run twice (the ring / counted waits are deterministic)."""
    onet, gnet = make_pair(ctx, 256, L, 32, S, S, 18, S * S + 1, 2)
    x = rand_planes(B, 18, S, S, seed=11)
    gnet.set_compute_mode(A.capi.COMPUTE_WINO_H2 | A.capi.COMPUTE_FORCE)
    gnet.set_wino_h2_gemm(1)
    p1, v1 = gnet.infer(x)
    gnet.set_wino_h2_gemm(2)
    p2, v2 = gnet.infer(x)
    p3, v3 = gnet.infer(x)
    gnet.set_wino_h2_gemm(1 + 64)                          # round 4's store policy (A/B hook): the same bits
    p4, v4 = gnet.infer(x)
    np.testing.assert_array_equal(p4, p1)
    np.testing.assert_array_equal(p2, p1)
    np.testing.assert_array_equal(v2, v1)
    np.testing.assert_array_equal(p3, p2)
    nb = min(B, 4)
    po, vo = onet.infer(x[:nb])
    np.testing.assert_allclose(p2[:nb], po, atol=POL_ATOL, rtol=POL_RTOL)
    np.testing.assert_allclose(v2[:nb], vo, atol=VAL_ATOL)
    with pytest.raises(A.AgzError):
        gnet.set_wino_h2_gemm(5)
    gnet.close()


def _heterogeneous_pair(ctx, K, L, W, H, F, Aspace, E, seed=5):
    """A network whose function equals a tame random network's, written with wildly different magnitudes INSIDE every layer:
    output channel c of every layer (both branches of a dual block) carries the scale s_c = 2^u, u ~ U[-E, E]; its filters have
    norms a_c = 2^u' (independently log-uniform, per branch), gamma makes up the rest (s_c / a_c, exact powers of two), beta is scaled
    by s_c, and whatever consumes channel c (the next block's filters, the head convolutions) is scaled by 1 / s_c.  ReLU commutes
    with positive scales, so only the rounding differs from the tame network — activations and weights now span 2^(2E) within one
    board / one layer, which is what the per-board / per-layer power-of-two ranges of the fp16x2 modes have to live with."""
    onet, gnet = make_pair(ctx, K, L, 32, W, H, F, Aspace, 2, seed=seed)
    rng = np.random.default_rng(seed + 77)
    HW = W * H
    names = [onet.param_name(i) for i in range(onet.num_params())]
    P = {n: onet.get_param(i).astype(np.float64).copy() for i, n in enumerate(names)}

    def pow2(n):
        return np.exp2(rng.integers(-E, E + 1, size=n).astype(np.float64))

    s_prev = None
    layers = [("FilterInit", "Init_gamma", "Init_beta", None, None, None)]
    for l in range(L):
        layers.append(("FilterLayer1 of Shared Layer %d" % l, "L1_%d_gamma" % l, "L1_%d_beta" % l,
                       "FilterLayer2 of Shared Layer %d" % l, "L2_%d_gamma" % l, "L2_%d_beta" % l))
    for (fa, ga, ba, fb, gb, bb) in layers:
        s = pow2(K)
        for (f, g, b_) in ((fa, ga, ba), (fb, gb, bb)):
            if f is None:
                continue
            cin = P[f].size // (K * 9)
            w = P[f].reshape(K, cin, 9)
            a = pow2(K)
            w *= a[:, None, None]
            if s_prev is not None:
                w /= s_prev[None, :, None]
            P[f] = w.reshape(-1)
            P[g] = (P[g].reshape(K, HW) * (s / a)[:, None]).reshape(-1)
            P[b_] = (P[b_].reshape(K, HW) * s[:, None]).reshape(-1)
        s_prev = s
    for f in ("FilterPolicyHead", "FilterValueHead"):
        w = P[f].reshape(-1, K)
        P[f] = (w / s_prev[None, :]).reshape(-1)
    for i, n in enumerate(names):
        v = P[n].astype(np.float32)
        onet.set_param(i, v)
        gnet.set_param(i, v)
    gnet.commit()
    return onet, gnet


@pytest.mark.parametrize("E", [4, 8, 12])
@pytest.mark.parametrize("wmode", [A.capi.COMPUTE_WINO_H2, A.capi.COMPUTE_FP16X2])
def test_fp16x2_modes_with_heterogeneous_ranges_inside_a_layer_k256(ctx, wmode, E):
    """channels (and filters) whose magnitudes differ by up to 2^(2E) inside one board / one layer, K = 256,
    19x19, against the oracle with the tolerance of every other network test.  AGZ_COMPUTE_WINO_H2 equilibrates every layer by
    exact powers of two (per input channel and per GEMM column, conv_wino_h2.hpp wino_build_u2) and has to hold it at every E;
    AGZ_COMPUTE_FP16X2 (one range per board / per layer, documented as such) is held to it at E = 4."""
    K, L, W, H, F, Aspace, B = 256, 2, 19, 19, 18, 362, 6
    onet, gnet = _heterogeneous_pair(ctx, K, L, W, H, F, Aspace, E)
    x = rand_planes(B, F, H, W, seed=E)
    pol_f, val_f = gnet.infer(x)                       # AGZ_COMPUTE_F32_MFMA, latency regime: exact fp32 products, split-K
    gnet.set_compute_mode(A.capi.COMPUTE_AUTO)
    pol_l, val_l = gnet.infer(x)                       # any other mode, latency regime: conv_lat.hpp's fp16x2 kernel
    gnet.set_compute_mode(wmode | A.capi.COMPUTE_FORCE)
    pol_g, val_g = gnet.infer(x)
    idx = [0, B - 1]
    pol_o, val_o = onet.infer(x[idx])
    print("\n[heterogeneous ranges] mode=%d E=%d: max|dpol| vs oracle %.3e (f32 path %.3e), vs f32 path %.3e; max|dval| %.3e; max p %.3e"
          % (wmode, E, np.abs(pol_g[idx] - pol_o).max(), np.abs(pol_f[idx] - pol_o).max(), np.abs(pol_g - pol_f).max(),
             np.abs(val_g[idx] - val_o).max(), pol_o.max()))
    assert np.all(np.isfinite(pol_g)) and np.all(np.isfinite(val_g))
    assert pol_o.max() < 0.9, "the test network saturated: parity on a one-hot policy would be vacuous"
    # this batch size is the latency regime.  The default mode keeps exact fp32 products there (agz.h promises them for
    # AGZ_COMPUTE_F32_MFMA at every batch size); every other mode takes conv_lat.hpp's fp16x2 kernel with ITS per-channel /
    # per-column equilibration and per-board ranges — both held to the full tolerance at every E, and they are different arithmetic
    np.testing.assert_allclose(pol_f[idx], pol_o, atol=POL_ATOL, rtol=POL_RTOL)
    np.testing.assert_allclose(val_f[idx], val_o, atol=VAL_ATOL)
    np.testing.assert_allclose(pol_l[idx], pol_o, atol=POL_ATOL, rtol=POL_RTOL)
    np.testing.assert_allclose(val_l[idx], val_o, atol=VAL_ATOL)
    assert not np.array_equal(pol_l, pol_f), "AGZ_COMPUTE_AUTO at batch 6 did not take the fp16x2 latency kernel"
    if wmode == A.capi.COMPUTE_FP16X2 and E > 4:
        # the direct fp16x2 mode keeps one range per board and one per layer (include/agz.h: elements more than 2^17 below their
        # board's / layer's maximum lose relative precision — opt-in mode): beyond E = 4 only finiteness is promised
        return
    np.testing.assert_allclose(pol_g[idx], pol_o, atol=POL_ATOL, rtol=POL_RTOL)
    np.testing.assert_allclose(val_g[idx], val_o, atol=VAL_ATOL)
