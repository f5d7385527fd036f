"""The error bound the screened top-k (csrc/topk_screen.hip) rests on, checked on the CPU in numpy:

    |s - s^| <= ||du|| ||i^|| + ||u|| (||di|| + 4e-5 ||i||),   u^ = bf16(u), du = u^ - u (likewise i),   s^ = u^ . i^ in fp32,

for the exact product s (float64), the fp32 rescoring s_r the merge kernel sorts by, and the quantities as the kernels form them
(norms in fp32 scaled by 1.00001 and rounded UP to bf16) — random rows of several scales and distributions, and adversarial rows
whose roundings all push the same way (r06: this test found the first version's bound, 1.03 x 2^-8 ||u|| ||i||, a factor two short —
round-to-nearest bf16 is off by up to 2^-8 per operand, not 2^-9; heavy-tailed rows exceeded it).  Also the pre-pass's packed
maxima: clearing / replacing the low 8 mantissa bits moves a value by < 2^-15 of itself, which its wider slack (+ 6.2e-5 ||u|| ||i||)
covers, and the threshold's lowering by 1e-5 |tau|.  No GPU: this pins the arithmetic the HIP kernels implement, not the kernels."""
import numpy as np
import pytest

FP_SLACK = np.float32(4.0e-5)
PACK_SLACK = np.float32(6.2e-5)


def bf16_rne(x):
    """float32 -> the nearest bf16 (ties to even), returned as float32."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def bf16_up(x):
    """smallest bf16 >= x (x >= 0), as float32 — bf16_up() of the kernels."""
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0xFFFF) >> 16) << 16).astype(np.uint32).view(np.float32)


def screen(u, it):
    """(s^, m^, m^_pre) as the kernels form them, fp32 throughout."""
    u, it = u.astype(np.float32), it.astype(np.float32)
    uh, ih = bf16_rne(u), bf16_rne(it)
    s_hat = np.zeros((u.shape[0], it.shape[0]), dtype=np.float32)
    for k in range(u.shape[1]):  # a fixed fp32 accumulation order (the matrix core's own order is covered by the slack)
        s_hat += uh[:, k:k + 1] * ih[None, :, k]
    one = np.float32(1.00001)
    norm = lambda x: np.sqrt((x * x).sum(1, dtype=np.float32))  # noqa: E731
    a_u, b_u = bf16_up(norm(uh - u) * one), bf16_up(norm(u) * one)
    ni = norm(it) * one
    c_i = bf16_up(norm(ih) * one)
    d_i = bf16_up(norm(ih - it) * one + FP_SLACK * ni)
    p_i = bf16_up(norm(ih - it) * one + (FP_SLACK + PACK_SLACK) * ni)
    m_hat = a_u[:, None] * c_i[None, :] + b_u[:, None] * d_i[None, :]
    m_pre = a_u[:, None] * c_i[None, :] + b_u[:, None] * p_i[None, :]
    return s_hat, m_hat, m_pre


def rescoring(u, it):
    """the merge kernel's exact score: fp32 FMAs, 16 lanes x 4 columns then a butterfly — any fp32 order is inside the slack."""
    return (u.astype(np.float32)[:, None, :] * it.astype(np.float32)[None, :, :]).sum(-1, dtype=np.float32)


def cases(rng, d):
    yield "normal", rng.standard_normal((64, d)), rng.standard_normal((96, d))
    yield "tiny", rng.standard_normal((64, d)) * 1e-20, rng.standard_normal((96, d)) * 1e-15
    yield "large", rng.standard_normal((64, d)) * 1e12, rng.standard_normal((96, d)) * 1e9
    yield "heavy tails", rng.standard_cauchy((64, d)), rng.standard_cauchy((96, d))
    yield "positive (no cancellation)", np.abs(rng.standard_normal((64, d))) + 0.5, np.abs(rng.standard_normal((96, d))) + 0.5
    # adversarial: every element sits just below a bf16 rounding boundary (relative error ~ 2^-9 each, all of one sign)
    base = bf16_rne(np.abs(rng.standard_normal((64, d))).astype(np.float32) + 1.0)
    ulp = np.float32(2.0) ** (np.floor(np.log2(base)) - 7)
    adv_u = (base + ulp * np.float32(0.499)).astype(np.float32)
    base_i = bf16_rne(np.abs(rng.standard_normal((96, d))).astype(np.float32) + 1.0)
    ulp_i = np.float32(2.0) ** (np.floor(np.log2(base_i)) - 7)
    yield "adversarial roundings", adv_u, (base_i + ulp_i * np.float32(0.499)).astype(np.float32)
    yield "one hot", np.eye(64, d) * 3.7, np.eye(96, d) * 0.9


@pytest.mark.parametrize("d", [64, 33, 128])
def test_the_screens_margin_bounds_what_bf16_loses(d):
    rng = np.random.default_rng(d)
    for name, u, it in cases(rng, d):
        u, it = u.astype(np.float32), it.astype(np.float32)
        s_hat, m_hat, _ = screen(u, it)
        s_true = u.astype(np.float64) @ it.astype(np.float64).T
        s_r = rescoring(u, it)
        slack = m_hat.astype(np.float64) - np.abs(s_true - s_hat.astype(np.float64))
        assert (slack >= 0).all(), (name, float(slack.min()))
        slack_r = m_hat.astype(np.float64) - np.abs(s_r.astype(np.float64) - s_hat.astype(np.float64))
        assert (slack_r >= 0).all(), (name, float(slack_r.min()))
        # the bound is not vacuous: on rows whose roundings all push one way the worst pair uses a good part of it, and it is well
        # inside the worst case of round-to-nearest bf16 (2^-7 ||u|| ||i||) on ordinary rows
        if name == "adversarial roundings":
            used = np.abs(s_true - s_hat) / m_hat
            assert used.max() > 0.4, float(used.max())
        if name == "normal":
            nn = np.linalg.norm(u, axis=1)[:, None] * np.linalg.norm(it, axis=1)[None, :]
            assert float((m_hat / nn).mean()) < 0.6 * 2.0 ** -7


def test_no_candidate_is_dropped_and_the_prepass_bound_is_a_lower_bound():
    """The two tests the kernels make: a pair passes the main pass when s^ + m^ - tau' > 0 with tau' = tau lowered by 1e-5 |tau|;
    the pre-pass keeps s^ - m^_pre with its low 8 mantissa bits replaced by a tile index.  With tau the k-th largest packed
    pre-pass value over a sample, every item whose rescoring score reaches the true k-th best passes."""
    rng = np.random.default_rng(7)
    d, k = 64, 10
    u = rng.standard_normal((32, d)).astype(np.float32) * np.float32(0.3)
    it = rng.standard_normal((4000, d)).astype(np.float32) * np.float32(0.3)
    s_hat, m_hat, m_pre = screen(u, it)
    lb = (s_hat - m_pre).astype(np.float32)
    packed = ((lb.view(np.uint32) & np.uint32(0xFFFFFF00)) | np.uint32(0xAB)).view(np.float32)     # any tile index in the low bits
    unpacked = (packed.view(np.uint32) & np.uint32(0xFFFFFF00)).view(np.float32)                  # what the threshold kernel reads
    s_r = rescoring(u, it)
    assert (unpacked.astype(np.float64) <= s_r.astype(np.float64)).all()                         # still a lower bound of the exact score
    sample = unpacked[:, :1000]
    tau = np.sort(sample, axis=1)[:, -k]                                                           # k-th largest lower bound of the sample
    kth = np.sort(s_r, axis=1)[:, -k]                                                              # the true k-th best
    assert (tau <= kth).all()
    t_low = tau - (np.abs(tau) * np.float32(1e-5) + np.float32(1e-30))
    passes = (s_hat + m_hat - t_low[:, None]) > 0
    needed = s_r >= kth[:, None]
    assert (passes | ~needed).all()
    # and the screen is selective: a few times k x n / sample candidates per user, not the whole table
    assert passes.sum(1).mean() < 20 * k
