"""CPU model of the index arithmetic of the latency regime's one-launch input layer (agogo_amd/csrc/conv_lat.hpp: lat_input_kernel):
the LDS-DMA of the workgroup's 288 x 64 weights (16 bytes per lane, four rows of 256 bytes per instruction, 18 instructions per wave),
the window of 8 pixels x 9 taps x 32 planes, and the range word index — restated in numpy against their definitions."""
import numpy as np
import pytest


@pytest.mark.parametrize("Kp,cg", [(256, 0), (256, 3), (128, 1), (64, 0)])
def test_weight_dma_lands_rows_of_the_channel_group(Kp, cg):
    wl = np.full((288, 64), -1, np.int64)                    # value = element index in wt[tap * 32 + plane][channel]
    for wid in range(4):
        for i in range(18):
            row0 = wid * 72 + i * 4
            for lane in range(64):
                vo = ((lane >> 4) * Kp + cg * 64 + (lane & 15) * 4) * 4          # lane's byte offset
                so = row0 * Kp * 4                                               # instruction's scalar offset
                src = (vo + so) // 4                                             # first of four floats
                dst_row, dst_col = row0 + (lane * 16) // 256, ((lane * 16) % 256) // 4   # lane-linear LDS destination at &wl[row0][0]
                for e in range(4):
                    assert wl[dst_row, dst_col + e] == -1
                    wl[dst_row, dst_col + e] = src + e
    rows, cols = np.meshgrid(np.arange(288), np.arange(64), indexing="ij")
    assert (wl == rows * Kp + cg * 64 + cols).all()


@pytest.mark.parametrize("W,H", [(19, 19), (9, 9), (7, 6)])
def test_window_fill_and_groups(W, H):
    HW, Hp, Wp = H * W, H + 2, W + 2
    groups = -(-HW // 8)
    covered = set()
    for grp in range(groups):
        p0 = grp * 8
        for e in range(8 * 72):                               # threads tid + 256 k over 8 pixels x 9 taps x 8 float4
            px, r = divmod(e, 72)
            tap, c4 = r >> 3, (r & 7) << 2
            p = p0 + px
            if p < HW:
                h, w = divmod(p, W)
                ky, kx = divmod(tap, 3)
                src_pix = (h + ky) * Wp + (w + kx)            # padded input: pixel (h, w) sits at (h + 1, w + 1)
                want = (h + 1 + ky - 1) * Wp + (w + 1 + kx - 1)
                assert src_pix == want and 0 <= src_pix < Hp * Wp and c4 + 3 < 32
                covered.add(p)
    assert covered == set(range(HW))
    for Kp in (64, 128, 256):
        words = groups * (Kp // 64)
        idx = {(grp * (Kp // 64) + cg) for grp in range(groups) for cg in range(Kp // 64)}
        assert idx == set(range(words))                       # words[b][group][channel group]: dense, one per workgroup
