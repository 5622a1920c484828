"""CPU model of the row arithmetic of the trainer's three-tap weight gradient (agogo_amd/csrc/train.hip: k_wgrad_h2t3).

The kernel's K steps are 32 consecutive valid rows of ONE board; the x operand lives in LDS as an image indexed by PADDED PIXEL (40
pixels starting at po(first row) + (ky - 1) * Wp - 1), and the row of the image that pairs with step row k under tap kx is
delta(k) + kx, delta(k) = k + 2 * (row ends between the step's first row and row k) — valid when a step crosses at most two row ends,
i.e. on boards >= 16 wide (the host's dispatch condition).  This test restates that arithmetic in numpy and checks, for every board
shape the dispatch admits around the sizes that occur (16..23 wide, several heights) and every step and tap, that it addresses exactly
the padded pixel the convolution's definition asks for, stays inside the 40-pixel image, and that the image never reads before the
buffer (negative pixels are masked in the kernel)."""
import numpy as np
import pytest


def delta(k, w0, W):
    c = w0 + k
    return k + (2 if c >= W else 0) + (2 if c >= 2 * W else 0)


@pytest.mark.parametrize("W", [16, 17, 19, 21, 23])
@pytest.mark.parametrize("H", [3, 16, 19, 22])
def test_three_tap_rows_address_the_convolutions_pixels(W, H):
    Hp, Wp, HW = H + 2, W + 2, H * W
    steps = -(-HW // 32)
    for b in (0, 3):
        for sib in range(steps):
            r0 = sib * 32
            h0, w0 = divmod(r0, W)
            po0 = (b * Hp + h0 + 1) * Wp + w0 + 1
            nv = min(32, HW - r0)
            for ky in range(3):
                qb = po0 + (ky - 1) * Wp - 1           # image row rho <-> padded pixel qb + rho
                for k in range(nv):
                    h, w = divmod(r0 + k, W)
                    po = (b * Hp + h + 1) * Wp + w + 1  # padded pixel of the step's row k (pix_off)
                    assert po == po0 + delta(k, w0, W)
                    for kx in range(3):
                        want = po + (ky - 1) * Wp + (kx - 1)   # x[pix(r) + off(tap)]: the convolution's definition
                        rho = delta(k, w0, W) + kx
                        assert 0 <= rho < 40
                        assert qb + rho == want
                        assert want >= 0                       # (the halo keeps every tap inside the buffer)


def test_boards_narrower_than_16_can_cross_three_row_ends_in_a_step():
    """why the dispatch keeps the register-staged kernel there: 32 rows of a 15-wide board can cross three row ends"""
    W = 15
    worst = max((w0 + 31) // W for w0 in range(W))
    assert worst == 3
    assert max((w0 + 31) // 16 for w0 in range(16)) == 2
