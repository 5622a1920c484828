"""CPU model of the LDS image and transposing-read addresses of the trainer's weight gradient (train.hip: k_wgrad_h2t3).

Hardware fact the kernel rests on (measured by scripts/probes/tr16_probe.hip on gfx950): `ds_read_b64_tr_b16` — every lane supplies the
address of 8 bytes (four halves); within a group of 16 lanes, result[lane c][j] = piece[j * 4 + c // 4][c % 4].  This test restates the
kernel's DMA fill (lane -> 16 bytes of the image) and its read addresses in numpy and checks that a lane's eight halves are exactly the
MFMA 32x32x16 operand fragment: column col0 + (lane & 31), k = 16 ks + 8 (lane >> 5) + e — for the dz image (rows = the step's k) and
for the x image (rows = padded pixels, row of (k, kx) = delta(k) + kx)."""
import numpy as np
import pytest

PA, PB = 32 * 128 * 2, 40 * 128 * 2


def tr16(lds, addr):
    """the transposing read: addr[64] byte addresses -> [64][4] halves"""
    piece = np.stack([lds[a // 2: a // 2 + 4] for a in addr])          # [64][4]
    out = np.empty((64, 4), lds.dtype)
    for l in range(64):
        g, c = l & ~15, l & 15
        for j in range(4):
            out[l, j] = piece[g + j * 4 + c // 4][c % 4]
    return out


def fill_dz(tile):
    """tile [32 k][128 col] -> image bytes as the DMA lands them: wave w, instruction t: rows 8 w + 4 t + (lane & 3), channels 8 (lane >> 2) .. + 7"""
    lds = np.zeros(PA // 2, np.int32)
    for w in range(4):
        for t in range(2):
            base = (2 * w + t) * 1024
            for lane in range(64):
                cs, kk = lane >> 2, lane & 3
                k = 8 * w + 4 * t + kk
                lds[(base + lane * 16) // 2: (base + lane * 16) // 2 + 8] = tile[k, 8 * cs: 8 * cs + 8]
    return lds


def lane_parts(lane):
    i16, kh = lane & 15, lane >> 5
    cpart = ((((lane >> 4) & 1) * 2 + ((i16 & 3) >> 1)) * 4) * 16 + (i16 & 1) * 8
    return i16, kh, cpart


def test_dz_fragment_addresses():
    tile = (np.arange(32)[:, None] * 1000 + np.arange(128)[None, :]).astype(np.int32)   # value = 1000 k + col
    lds = fill_dz(tile)
    for col0 in (0, 32, 64, 96):
        for ks in range(2):
            a0 = np.empty(64, np.int64)
            for lane in range(64):
                i16, kh, cpart = lane_parts(lane)
                a0[lane] = kh * 2048 + (i16 >> 2) * 16 + cpart + col0 * 8 + ks * 4096
            v = np.concatenate([tr16(lds, a0), tr16(lds, a0 + 1024)], axis=1)   # [64][8]
            for lane in range(64):
                for e in range(8):
                    assert v[lane, e] == 1000 * (16 * ks + 8 * (lane >> 5) + e) + col0 + (lane & 31)


def delta(k, w0, W):
    c = w0 + k
    return k + (2 if c >= W else 0) + (2 if c >= 2 * W else 0)


@pytest.mark.parametrize("w0", [0, 7, 13, 18])
def test_x_fragment_addresses(w0):
    W = 19
    img = (np.arange(40)[:, None] * 1000 + np.arange(128)[None, :]).astype(np.int32)    # value = 1000 rho + col
    lds = np.zeros(PB // 2, np.int32)
    for m in range(10):                               # ten four-row groups, lane -> row 4 m + (lane & 3), channels 8 (lane >> 2) ..
        for lane in range(64):
            cs, kk = lane >> 2, lane & 3
            lds[(m * 1024 + lane * 16) // 2: (m * 1024 + lane * 16) // 2 + 8] = img[4 * m + kk, 8 * cs: 8 * cs + 8]
    for col0 in (0, 64):
        for ks in range(2):
            for kx in range(3):
                a0, a1 = np.empty(64, np.int64), np.empty(64, np.int64)
                for lane in range(64):
                    i16, kh, cpart = lane_parts(lane)
                    k0 = ks * 16 + kh * 8 + (i16 >> 2)
                    r0, r1 = delta(k0, w0, W) + kx, delta(k0 + 4, w0, W) + kx
                    a0[lane] = (r0 >> 2) * 1024 + (r0 & 3) * 16 + cpart + col0 * 8
                    a1[lane] = (r1 >> 2) * 1024 + (r1 & 3) * 16 + cpart + col0 * 8
                v = np.concatenate([tr16(lds, a0), tr16(lds, a1)], axis=1)
                for lane in range(64):
                    for e in range(8):
                        k = 16 * ks + 8 * (lane >> 5) + e
                        assert v[lane, e] == 1000 * (delta(k, w0, W) + kx) + col0 + (lane & 31)


def _worst_way(addr):
    from collections import Counter
    return max(max(Counter((addr[l] % 256) // 8 for l in range(32 * h, 32 * h + 32)).values()) for h in (0, 1))


def test_transposing_reads_and_lds_banks():
    """a transposing read serves 32 lanes per LDS cycle (8 bytes each: one 256-byte bank row when every lane hits its own 8-byte slot).
    dz image: always 32 distinct slots.  x image: the rows of a 16-lane group are delta(k) + kx for four consecutive k — consecutive
    rows except across a row end (+2), where two lanes can meet on a slot: at most two-way, on under a third of the reads."""
    for ks in range(2):
        for e in range(2):
            addr = []
            for lane in range(64):
                i16, kh, cpart = lane_parts(lane)
                addr.append(kh * 2048 + (i16 >> 2) * 16 + cpart + ks * 4096 + e * 1024)
            assert _worst_way(addr) == 1
    W, total, conflicting = 19, 0, 0
    for w0 in range(W):
        for ks in range(2):
            for kx in range(3):
                for e in range(2):
                    addr = []
                    for lane in range(64):
                        i16, kh, cpart = lane_parts(lane)
                        r = delta(ks * 16 + kh * 8 + (i16 >> 2) + 4 * e, w0, W) + kx
                        addr.append((r >> 2) * 1024 + (r & 3) * 16 + cpart)
                    way = _worst_way(addr)
                    assert way <= 2
                    total += 1
                    conflicting += way > 1
    assert conflicting * 3 < total
