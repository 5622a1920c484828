"""CPU: the numpy restatement of Winograd F(4x4,3x3) that tests/test_wino_gpu.py checks the device stages against is itself
pinned here against a direct 3x3 convolution (float64), on ragged boards and channel counts."""
import numpy as np
import pytest

from test_wino_gpu import AT, BT, G, direct_conv, numpy_stages


@pytest.mark.parametrize("B,H,W,C,N", [(2, 7, 5, 16, 9), (1, 4, 4, 3, 5), (2, 19, 19, 8, 4), (1, 1, 1, 2, 2), (1, 6, 9, 4, 3)])
def test_numpy_winograd_equals_direct_convolution(B, H, W, C, N):
    rng = np.random.default_rng(B + 10 * H + 100 * C)
    x = np.maximum(rng.normal(0, 1, (B, H, W, C)), 0)
    w = rng.uniform(-1, 1, (N, C, 3, 3)).astype(np.float32)
    V, M, y = numpy_stages(x, w)
    nt = ((H + 3) // 4) * ((W + 3) // 4)
    assert V.shape == (36, B * nt, C) and M.shape == (36, B * nt, N)
    d = direct_conv(x, w)
    # the only rounding in the model: U = G g Gt rounded once to float32, like the library's commit
    np.testing.assert_allclose(y, d, atol=2e-6 * max(np.abs(d).max(), 1e-30))


def test_transform_matrices_are_the_f43_set():
    """At (G g Gt . Bt d B) A = 1-D correlation for polynomial points 0, +-1, +-2, inf: check the 1-D identity exactly"""
    rng = np.random.default_rng(0)
    for _ in range(10):
        d = rng.integers(-8, 9, 6).astype(np.float64)
        g = rng.integers(-8, 9, 3).astype(np.float64)
        y = AT @ ((G @ g) * (BT @ d))
        ref = np.array([d[i] * g[0] + d[i + 1] * g[1] + d[i + 2] * g[2] for i in range(4)])
        np.testing.assert_allclose(y, ref, atol=1e-9)
