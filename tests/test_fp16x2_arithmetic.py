"""The arithmetic of the fused grad kernels' 64 x 64 products (csrc/mlp64x16.hip, Lds16 CH = 3, `split2_pair`,
`chain64_f2`), restated in NumPy so that its error claim is checked where there is no GPU: every fp32 operand,
scaled by a power of two to just below 2^14, is split into two binary16 terms (round to nearest, the residual
exact in fp32) and a product is the fp32-accumulated sum of three fp16 MFMAs per 32-wide K block (lo.hi, hi.lo,
hi.hi).  Against float64 the result must be as close as a plain fp32 product of the same operands (what
`v_mfma_f32_16x16x4_f32` gives, 4 k-values per instruction) — for activations of ordinary size, tiny ones, ones
spread over five decades, saturated ones — and the split itself leaves at most 2^-23 of the operand (23
significant bits in the worst case: hi 11, lo 11, and the sign of lo)."""
import numpy as np
import pytest


def split2(a):
    """a = hi + lo + r: hi = fp16(a), lo = fp16(a - hi) (v_cvt_pk_f16_f32, v_fma_mix_f32, v_cvt_pk_f16_f32)"""
    a = a.astype(np.float32)
    hi = a.astype(np.float16)
    residual = (a - hi.astype(np.float32)).astype(np.float32)          # exact in fp32
    assert np.array_equal(residual.astype(np.float64), a.astype(np.float64) - hi.astype(np.float64))
    return hi, residual.astype(np.float16)


def mfma_terms(pairs, block):
    """fp32 accumulator; one instruction = the exact sum of `block` products added with one rounding."""
    rows, cols, depth = pairs[0][0].shape[0], pairs[0][1].shape[1], pairs[0][0].shape[1]
    acc = np.zeros((rows, cols), np.float32)
    for k in range(0, depth, block):
        for a, b in pairs:
            acc = (acc.astype(np.float64)
                   + a[:, k:k + block].astype(np.float64) @ b[k:k + block].astype(np.float64)).astype(np.float32)
    return acc


def power_of_two_below(top, largest):
    return np.float32(2.0 ** (top - (np.frexp(largest)[1])))           # largest < 2^e  ->  scaled < 2^top


CASES = {
    'ordinary': lambda rng, s: np.tanh(rng.standard_normal(s)),
    'tiny': lambda rng, s: np.tanh(1e-3 * rng.standard_normal(s)),
    'five decades': lambda rng, s: np.tanh(rng.standard_normal(s) * 10.0 ** rng.uniform(-4, 1, s)),
    'saturated': lambda rng, s: np.tanh(5 * rng.standard_normal(s)),
}


@pytest.mark.parametrize('case', list(CASES))
def test_three_fp16_mfmas_per_product_are_fp32_class(case):
    rng = np.random.RandomState(5)
    weights = (rng.uniform(-1, 1, (64, 64)) * 0.3).astype(np.float32)
    h = CASES[case](rng, (64, 2048)).astype(np.float32)
    exact = weights.astype(np.float64) @ h.astype(np.float64)
    plain = mfma_terms([(weights, h)], 4)                              # 16x16x4 fp32 MFMAs
    w_scale, h_scale = power_of_two_below(14, np.abs(weights).max()), np.float32(2.0 ** 14)
    w_hi, w_lo = split2(weights * w_scale)
    h_hi, h_lo = split2(h * h_scale)
    assert np.abs(w_hi.astype(np.float32)).max() < 2.0 ** 14 and np.isfinite(h_hi.astype(np.float32)).all()
    product = mfma_terms([(w_lo, h_hi), (w_hi, h_lo), (w_hi, h_hi)], 32).astype(np.float64) / (
        float(w_scale) * float(h_scale))
    top = np.abs(exact).max()
    error_fp32, error_split = np.abs(plain - exact).max() / top, np.abs(product - exact).max() / top
    assert error_fp32 < 4e-7
    assert error_split <= 1.5 * error_fp32, (case, error_fp32, error_split)


def test_the_split_keeps_23_significant_bits():
    rng = np.random.RandomState(6)
    a = (rng.standard_normal(1 << 16) * 10.0 ** rng.uniform(-3, 0, 1 << 16)).astype(np.float32)
    a = a * power_of_two_below(14, np.abs(a).max())                    # the largest just below 2^14
    hi, lo = split2(a)
    left = np.abs(a.astype(np.float64) - hi.astype(np.float64) - lo.astype(np.float64))
    # from 2^-1 up the low term is a normal binary16 number or has all its bits above the subnormal grid:
    # |a - hi| <= 2^-11 |a| (half an ulp of 11 bits), |(a - hi) - lo| <= 2^-23 |a| ...
    big = np.abs(a) >= 0.5
    assert (left[big] <= 2.0 ** -23 * np.abs(a[big])).all()
    assert (left[big] > 2.0 ** -24 * np.abs(a[big])).any(), 'the bound is attained: this split is not exact'
    # ... and the binary16 subnormal grid (2^-24) bounds what is lost below: 2^-25 absolute = 2^-39 of the largest
    assert left[~big].max() <= 2.0 ** -25


def test_the_running_unit_never_lets_an_operand_leave_binary16():
    """`enter_unit`: a tile announces the exponent e of a bound on its values BEFORE they are scaled; the unit
    2^(14 - e_run) only shrinks (e_run grows, with two binades of headroom), accumulators are rescaled by the
    exact ratio.  Whatever the order of magnitudes, no scaled value reaches 2^14 and the sum comes out as if
    nothing had been scaled."""
    rng = np.random.RandomState(7)
    for order in ('rising', 'falling', 'shuffled'):
        decades = {'rising': np.linspace(-12, 6, 400), 'falling': np.linspace(6, -12, 400),
                   'shuffled': rng.uniform(-12, 6, 400)}[order]
        tiles = [(rng.standard_normal(16) * 10.0 ** d).astype(np.float32) for d in decades]
        e_run, accumulator, rescales = -100, np.float64(0.0), 0
        for tile in tiles:
            e = int(np.frexp(np.abs(tile).max())[1])
            if e > e_run:
                e_new = min(e + 2, 110)
                accumulator *= 2.0 ** (e_run - e_new)
                e_run, rescales = e_new, rescales + 1
            scaled = tile.astype(np.float64) * 2.0 ** (14 - e_run)
            assert np.abs(scaled).max() < 2.0 ** 14
            accumulator += scaled.sum()
        total = sum(t.astype(np.float64).sum() for t in tiles)
        assert abs(accumulator * 2.0 ** (e_run - 14) - total) <= 1e-12 * sum(np.abs(t).sum() for t in tiles)
        assert rescales <= (40 if order == 'rising' else 12), (order, rescales)


def layer_one(weights, x, equilibrate):
    """Layer 1 of the shipped grad kernels on fp16x2 terms (csrc/mlp64x16.hip, stage_weights16 / the tile loop):
    z1 = W1 x, each SAMPLE's inputs in the unit of its own largest entry.  equilibrate: column k of the image is
    W1[:, k] 2^(12 - e_k), e_k the exponent of max |W1[:, k]|, and input k enters times 2^(e_k) (Lds16::CX)."""
    cols = weights.shape[1]
    wp, xp = np.zeros((64, 32), np.float32), np.zeros((32, x.shape[0]), np.float32)
    wp[:, :cols], xp[:cols] = weights, x.T
    if equilibrate:
        top = np.abs(wp).max(0)
        e_k = np.clip(np.where(top > 0, np.frexp(top)[1], 0), -40, 40)
        image, inputs, unit = wp * np.float32(2.0) ** (12 - e_k)[None, :], xp * (np.float32(2.0) ** e_k)[:, None], 2.0 ** -12
    else:
        e = int(np.frexp(np.abs(wp).max())[1])
        image, inputs, unit = wp * np.float32(2.0 ** (12 - e)), xp, 2.0 ** (e - 12)
    ex = np.clip(np.frexp(np.abs(inputs).max(0))[1], -38, 100)
    sx = np.float32(2.0) ** (14 - ex)
    x_hi, x_lo = split2(inputs * sx[None, :])
    w_hi, w_lo = split2(image)
    assert np.isfinite(x_hi.astype(np.float32)).all() and np.isfinite(w_hi.astype(np.float32)).all()
    return mfma_terms([(w_lo, x_hi), (w_hi, x_lo), (w_hi, x_hi)], 32).astype(np.float64) * unit / sx[None, :].astype(np.float64)


@pytest.mark.parametrize('outlier', [False, True])
def test_layer_one_is_fp32_class_for_features_seven_decades_apart(outlier):
    """The reference's actor sees raw observations (models/actors.py:128-129): columns 1e-3 ... 1e4 with weights
    that undo the scales (every term of z1 is O(1)), with and without a single 1e6 entry.  In the unit of the
    sample's largest |x| alone, a feature 2^-17 below it loses its low term to binary16's subnormal grid: the
    error is 1e-4 of sum |w||x|; with the columns equilibrated it is the fp32 MFMA's, x 2 at most."""
    rng = np.random.RandomState(3)
    cols, n = 17, 2048
    scales = 10.0 ** np.linspace(-3, 4, cols)
    rng.shuffle(scales)
    x = (rng.standard_normal((n, cols)) * scales).astype(np.float32)
    if outlier:
        x[7, 3] = 1e6
    weights = (rng.normal(size=(64, cols)) * 0.3 / scales).astype(np.float32)
    exact = weights.astype(np.float64) @ x.T.astype(np.float64)
    size = np.abs(weights).astype(np.float64) @ np.abs(x.T).astype(np.float64)
    wp, xp = np.zeros((64, 32), np.float32), np.zeros((32, n), np.float32)
    wp[:, :cols], xp[:cols] = weights, x.T
    fp32 = (np.abs(mfma_terms([(wp, xp)], 4) - exact) / size).max()
    unit_only = (np.abs(layer_one(weights, x, False) - exact) / size).max()
    shipped = (np.abs(layer_one(weights, x, True) - exact) / size).max()
    assert fp32 < 3e-7
    assert unit_only > 1e-5, 'the case no longer shows what the equilibration is for'
    assert shipped <= 2.0 * fp32, (fp32, shipped)


def test_quotient_by_reciprocal_and_one_newton_step_is_the_division():
    """`quotient_by` (csrc/mlp64x16.hip): (x - mean) / std of mean_stds.py:36 as q0 = c r, q = fma(fma(-q0, sd, c), r, q0)
    with r = fl(1 / sd).  Over 4 M operand pairs around the std floor the result is the IEEE quotient (the plain
    product c r differs from it in ~25 % of the cases)."""
    rng = np.random.RandomState(9)
    c = (rng.standard_normal(1 << 22) * 10.0 ** rng.uniform(-3, 3, 1 << 22)).astype(np.float32)
    sd = np.maximum(10.0 ** rng.uniform(-2, 2, 1 << 22), 1e-2).astype(np.float32)
    r = (np.float32(1) / sd).astype(np.float32)
    q0 = (c * r).astype(np.float32)
    L = np.longdouble
    rem = (c.astype(L) - q0.astype(L) * sd.astype(L)).astype(np.float32)          # exact in the FMA, then rounded
    q = (q0.astype(L) + rem.astype(L) * r.astype(L)).astype(np.float32)
    want = (c / sd).astype(np.float32)
    assert (q0 != want).mean() > 0.05
    assert np.array_equal(q, want)
