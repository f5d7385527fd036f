"""The arithmetic of the fp16 two-term operands (csrc/mfma_common.h split2_f16, csrc/lse.hip option "lse_f16"), pinned in numpy on
the CPU: numpy's float16 conversion is IEEE round-to-nearest-even with gradual underflow, as v_cvt_pk_f16_f32 is.

What DESIGN.md §2.4b claims and the kernels rely on:
  * x * scale = h + l up to 2^-22 |x * scale| whenever l is a normal fp16 number, and up to 2^-25 absolutely when it is not;
  * the three products h h' + h l' + l h' reproduce x x' to ~ 3 x 2^-22 (the dropped l l' is <= 2^-22);
  * unit rows scaled by 2^8 and weights in [0, 1] scaled by 2^14 never overflow fp16 (65 504);
  * a dot product of unit rows formed that way is within 1e-6 of float64 — the exponent's error budget at tau = 0.05."""
import numpy as np

ROW_SCALE, W_SCALE = 256.0, 16384.0


def split2(x, scale):
    v = (np.asarray(x, dtype=np.float32) * np.float32(scale)).astype(np.float32)
    h = v.astype(np.float16)
    r = (v - h.astype(np.float32)).astype(np.float32)  # exact in fp32
    l = r.astype(np.float16)
    return h, l, v


def test_two_terms_carry_22_bits():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-1, 1, 200000), rng.uniform(-1, 1, 50000) * 10.0 ** rng.uniform(-8, 0, 50000),
                        [1.0, -1.0, 0.0, 2.0 ** -11, 2.0 ** -14, 1 - 2.0 ** -24]]).astype(np.float32)
    h, l, v = split2(x, ROW_SCALE)
    assert np.all(np.isfinite(h.astype(np.float32))) and float(np.abs(h.astype(np.float32)).max()) <= 256.0
    err = np.abs(v.astype(np.float64) - (h.astype(np.float64) + l.astype(np.float64)))
    normal_l = np.abs(l.astype(np.float64)) >= 2.0 ** -14
    assert np.all(err[normal_l] <= 2.0 ** -22 * np.abs(v[normal_l].astype(np.float64)))
    assert np.all(err[~normal_l] <= 2.0 ** -25)  # half a subnormal step: 1.2e-10 of the UNscaled value
    # (a low term below 2^-14 is either the residual of an element below 2^-3 / scale or a residual that happens to be tiny: both
    #  are covered by the absolute bound)


def test_weights_scaled_by_2_to_14_stay_in_range():
    w = np.concatenate([np.linspace(0, 1, 10001), [1.0 + 2.0 ** -20, 4.5e-5, 2e-9, 1e-12]]).astype(np.float32)
    h, l, v = split2(w, W_SCALE)
    assert np.all(np.isfinite(h.astype(np.float32))) and float(h.astype(np.float32).max()) <= 16400.0 < 65504.0
    err = np.abs(v.astype(np.float64) - (h.astype(np.float64) + l.astype(np.float64)))
    assert np.all(err <= np.maximum(2.0 ** -22 * v.astype(np.float64), 2.0 ** -25))


def test_three_products_on_unit_rows():
    rng = np.random.default_rng(1)
    for d in (64, 128):
        a = rng.standard_normal((512, d)) * np.exp(2.0 * rng.standard_normal((512, d)))  # heavy tails inside a row
        b = rng.standard_normal((512, d)) * np.exp(2.0 * rng.standard_normal((512, d)))
        a /= np.linalg.norm(a, axis=1, keepdims=True)
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        a32, b32 = a.astype(np.float32), b.astype(np.float32)
        ah, al, _ = split2(a32, ROW_SCALE)
        bh, bl, _ = split2(b32, ROW_SCALE)
        f = lambda t: t.astype(np.float64)  # noqa: E731  (the matrix core's products are exact, its fp32 sum is not the point here)
        got = (f(ah) @ f(bh).T + f(ah) @ f(bl).T + f(al) @ f(bh).T) / (ROW_SCALE * ROW_SCALE)
        ref = f(a32) @ f(b32).T
        # |x| <= 1 for unit rows: 3 x 2^-22 x sum |a_k b_k| <= 3 x 2^-22
        assert float(np.abs(got - ref).max()) <= 3.0 * 2.0 ** -22
        assert float(np.abs(got - ref).max()) <= 1e-6  # exp((x - 1) / 0.05) moves by < 2e-5 relative at this error


def test_second_product_weight_times_scaled_rows():
    rng = np.random.default_rng(2)
    p = rng.uniform(0, 1, (64, 256)).astype(np.float32) ** 8  # probabilities, most of them small
    c = rng.standard_normal((256, 64))
    c = (c / np.linalg.norm(c, axis=1, keepdims=True)).astype(np.float32)
    ph, pl, _ = split2(p, W_SCALE)
    ch, cl, _ = split2(c, ROW_SCALE)
    f = lambda t: t.astype(np.float64)  # noqa: E731
    got = (f(ph) @ f(ch) + f(ph) @ f(cl) + f(pl) @ f(ch)) / (W_SCALE * ROW_SCALE)
    ref = f(p) @ f(c)
    scale = np.abs(f(p)) @ np.abs(f(c))
    assert np.all(np.abs(got - ref) <= 3.0 * 2.0 ** -22 * scale + 1e-12)
