"""A numpy model of the fp16x2 product arithmetic with power-of-two equilibration (conv_lat.hpp, conv_wino_h2.hpp, train.hip k_wgrad_h2):
what the device kernels do, restated on the CPU so that the numerical claims of DESIGN.md 4a / 4c can be checked without a GPU:
  * x = hi + lo with hi = RN16(x s), lo = RN16(x s - hi) carries x s to ~2^-22 relative for elements within 2^17 of the range;
  * three products (hi hi, hi lo, lo hi) accumulated in fp32 reproduce a K = 2304 dot product as well as fp32 products do;
  * with channels spread over 2^(+-E) inside a layer, ONE scale per tensor loses the small channels, the per-input-channel /
    per-column powers of two (t_in, col_unscale) keep every channel at full precision — the failure tests/test_wino_gpu.py found on
    the device in round 3 and the fix it pins."""
import numpy as np


def split16(v):
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi, lo


def pow2_floor(a):
    """2^floor(log2 a) for a > 0 (exact)"""
    m, e = np.frexp(a)          # a = m 2^e, m in [0.5, 1)
    return np.ldexp(1.0, e - 1)


def range_scale(amax):
    """the power of two that puts amax into [2^13, 2^14)"""
    m, e = np.frexp(np.float64(amax))
    return np.ldexp(1.0, 14 - e)


def dot_fp16x2(x, w, sx, sw):
    """sum_k x_k w_k with both operands scaled (sx, sw: scalars or per-k vectors of powers of two), split hi/lo, three products, fp32 sum"""
    xh, xl = split16((x * sx).astype(np.float32))
    wh, wl = split16((w * sw).astype(np.float32))
    acc = np.float32(0)
    # (the MFMA accumulates exactly-representable fp16 x fp16 products in fp32)
    for a, b in ((xl, wh), (xh, wl), (xh, wh)):
        acc = np.float32(acc + np.sum(a.astype(np.float32) * b.astype(np.float32), dtype=np.float32))
    return acc


def test_hi_lo_split_carries_22_bits_within_range():
    rng = np.random.default_rng(1)
    x = (rng.normal(size=4000) * np.exp2(rng.integers(-10, 4, size=4000))).astype(np.float32)
    s = np.float32(range_scale(np.abs(x).max()))
    hi, lo = split16(x * s)
    xs = x.astype(np.float64) * float(s)
    back = hi.astype(np.float64) + lo.astype(np.float64)
    assert np.isfinite(hi.astype(np.float32)).all()          # max |x s| < 2^14: no overflow
    assert np.abs(back - xs).max() <= 2.0 ** -25 * 1.0001 + np.abs(xs).max() * 2.0 ** -23   # absolute: lo's grid (2^-24) / fp16's 11+11 bits
    near = np.abs(xs) >= 2.0 ** -3                            # within 2^17 of the range: lo is a normal fp16 number
    assert (np.abs(back - xs)[near] / np.abs(xs)[near]).max() < 2.0 ** -21


def test_three_products_match_fp32_products_on_a_k2304_dot():
    rng = np.random.default_rng(2)
    errs_h2, errs_f32 = [], []
    for _ in range(200):
        x = np.maximum(rng.normal(size=2304), 0).astype(np.float32)          # post-ReLU activations
        w = (rng.uniform(-1, 1, size=2304) * 0.03).astype(np.float32)          # Glorot-like filter row
        exact = np.dot(x.astype(np.float64), w.astype(np.float64))
        got = dot_fp16x2(x, w, np.float32(range_scale(np.abs(x).max())), np.float32(range_scale(np.abs(w).max())))
        got = got / (range_scale(np.abs(x).max()) * range_scale(np.abs(w).max()))
        f32 = np.sum(x * w, dtype=np.float32)
        scale = np.abs(x.astype(np.float64) * w.astype(np.float64)).sum()
        errs_h2.append(abs(got - exact) / scale)
        errs_f32.append(abs(np.float64(f32) - exact) / scale)
    assert np.max(errs_h2) < 4e-7
    assert np.median(errs_h2) < 3 * np.median(errs_f32) + 1e-9               # fp32 ACCUMULATION dominates both


def test_equilibration_keeps_heterogeneous_channels_exact_where_one_scale_fails():
    rng = np.random.default_rng(3)
    K, taps, E = 256, 9, 12
    u = np.exp2(rng.integers(-E, E + 1, size=K).astype(np.float64))           # channel c of the input carries the scale u_c ...
    x_tame = np.maximum(rng.normal(size=(K, taps)), 0)
    w_tame = rng.uniform(-1, 1, size=(K, taps)) * 0.03
    x = (x_tame * u[:, None]).astype(np.float32)                              # ... and the filter the scale 1 / u_c: same function
    w = (w_tame / u[:, None]).astype(np.float32)
    exact = np.sum(x.astype(np.float64) * w.astype(np.float64))
    mag = np.abs(x.astype(np.float64) * w.astype(np.float64)).sum()
    # (a) one power of two per tensor (round 2's form)
    one = dot_fp16x2(x.ravel(), w.ravel(), np.float32(range_scale(np.abs(x).max())), np.float32(range_scale(np.abs(w).max())))
    one = one / (range_scale(np.abs(x).max()) * range_scale(np.abs(w).max()))
    # (b) per-input-channel t_in from the weights' rows (inverse folded into the weights), then the tensors' ranges
    t_in = pow2_floor(np.abs(w).max(axis=1).astype(np.float64))
    xe = (x * t_in[:, None]).astype(np.float32)                               # exact: powers of two
    we = (w / t_in[:, None]).astype(np.float32)
    sx, sw = range_scale(np.abs(xe).max()), range_scale(np.abs(we).max())
    eq = dot_fp16x2(xe.ravel(), we.ravel(), np.float32(sx), np.float32(sw)) / (sx * sw)
    err_one, err_eq = abs(one - exact) / mag, abs(eq - exact) / mag
    assert err_eq < 1e-7, err_eq
    assert err_one > 10 * err_eq, (err_one, err_eq)   # one dot product: mild (~2e-7); through Winograd transforms and 20 layers the device
                                                        # test measured 2.9e-4 at E = 8 before the fix and 6e-9 after
