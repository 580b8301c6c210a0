"""The arithmetic behind the fp32 kernels on the bf16 matrix pipe (csrc/conv_common.hpp split3, conv_f32x3.hip,
k_wgrad_f32x3), restated in numpy: an fp32 value (|a| >= 2^-100, or 0) is the EXACT sum of three bf16 values, and the
six largest of the nine term products reproduce an fp32 x fp32 product to 2^-20 relative in the worst case and 2^-24
on average — the size of fp32 rounding errors, which is why those kernels are held to the same 1e-4 / fp32-grade
tests as the fp32-MFMA kernels (tests/test_gpu_conv.py).  Below 2^-100 the third term can fall into the bf16
subnormals and lose bits: an absolute error under 2^-133.  No GPU, no library: this pins the number-format argument."""
import numpy as np


def _trunc_bf16(a):
    """upper 16 bits of the fp32 encoding (a bf16 value held in fp32)"""
    return (a.astype(np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def _split3(a):
    a = a.astype(np.float32)
    a1 = _trunc_bf16(a)
    r = (a - a1).astype(np.float32)          # exact: the low 16 mantissa bits of a
    a2 = _trunc_bf16(r)
    r2 = (r - a2).astype(np.float32)         # exact: at most 8 significant bits are left
    a3 = _trunc_bf16(r2)
    return a1, a2, a3, r2


def _samples():
    g = np.random.default_rng(0)
    x = np.concatenate([
        g.standard_normal(200000).astype(np.float32),
        (g.standard_normal(50000) * 1e-20).astype(np.float32),
        (g.standard_normal(50000) * 1e20).astype(np.float32),
        np.float32([0.0, -0.0, 1.0, -1.0, 3.0, 1 + 2 ** -23, 1 - 2 ** -24, 2 ** -126, 2 ** -130, 65504.0, 1e-38]),
        g.integers(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.uint32).view(np.float32),
    ])
    return x[np.isfinite(x)]


def test_three_bf16_terms_are_exact():
    a = _samples()
    a1, a2, a3, r2 = _split3(a)
    for t in (a1, a2, a3):
        assert not np.any(t.view(np.uint32) & np.uint32(0xFFFF))   # each term is a bf16 value
    total = a1.astype(np.float64) + a2.astype(np.float64) + a3.astype(np.float64)
    big = (np.abs(a) >= 2.0 ** -100) | (a == 0)
    assert np.array_equal(a3[big], r2[big])                        # nothing is lost by the third truncation
    assert np.array_equal(total[big], a[big].astype(np.float64))
    # tiny values: the remainder can be a bf16 subnormal with bits below its 7 stored ones
    assert np.all(np.abs(total[~big] - a[~big].astype(np.float64)) < 2.0 ** -133)
    # magnitudes: every term is at least 2^-8 below the previous one (what orders the six products)
    nz = a1 != 0
    assert np.all(np.abs(a2[nz]) <= np.abs(a1[nz]) * 2.0 ** -7)
    assert np.all(np.abs(a3[nz]) <= np.abs(a1[nz]) * 2.0 ** -15)


def test_six_products_are_fp32_grade():
    g = np.random.default_rng(1)
    a = g.standard_normal(400000).astype(np.float32)
    w = (g.standard_normal(400000) * 0.3).astype(np.float32)
    sa, sw = _split3(a)[:3], _split3(w)[:3]
    # the kernels issue the small terms first into one fp32 accumulator chain; bf16 x bf16 products are exact in fp32,
    # here the six of them are summed in float64 to isolate what DROPPING the other three costs
    kept = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]
    got = sum(sa[i].astype(np.float64) * sw[j].astype(np.float64) for i, j in kept)
    exact = a.astype(np.float64) * w.astype(np.float64)
    rel = np.abs(got - exact) / np.maximum(np.abs(exact), 1e-300)
    # dropped: a2*w3 + a3*w2 + a3*w3 with |a2| < 2^-7 |a|, |a3| < 2^-14 |a|
    assert rel.max() <= 2.0 ** -20, rel.max()
    assert rel.mean() <= 2.0 ** -24, rel.mean()
    # a plain bf16 product (what the bf16 FEATURE path does by design) is 2^-8: fourteen bits worse
    rel1 = np.abs(sa[0].astype(np.float64) * sw[0].astype(np.float64) - exact) / np.maximum(np.abs(exact), 1e-300)
    assert rel1.max() > 2.0 ** -9


def _split3_planes(a, weight_side=False):
    """numpy restatement of split3's non-finite rule (csrc/conv_common.hpp): -> three bf16 planes held in fp32"""
    a = np.asarray(a, np.float32)
    with np.errstate(invalid="ignore"):
        a1, a2, a3, _ = _split3(np.where(np.isfinite(a), a, np.float32(0)))
    bits = a.view(np.uint32)
    t = ((bits & np.uint32(0xFFFF0000)) | np.where(bits & np.uint32(0x007FFFFF), np.uint32(0x00400000), np.uint32(0)))
    t = t.astype(np.uint32).view(np.float32)
    bad = ~np.isfinite(a)
    if weight_side:
        return [np.where(bad, np.float32(np.nan), p) for p in (a1, a2, a3)]
    return [np.where(bad, np.float32(0), a1), np.where(bad, np.float32(0), a2), np.where(bad, t, a3)]


def test_nonfinite_rule_matches_fp32_products():
    """The six kept products with a non-finite ROW value in plane 3 alone reproduce the IEEE class of a * w (sign of
    the infinity, NaN for NaN and for inf * 0) for every finite w with a non-zero leading plane — in particular for
    weights exact in bf16, whose second and third planes are zero (inf * 0 would poison the sum if the infinity sat in
    plane 1).  A non-finite weight-side value makes the product NaN; inf * inf is never silently dropped."""
    inf = np.float32(np.inf)
    a = np.float32([inf, -inf, np.nan, inf, -inf, inf, inf, 2.0, -3.0, 0.0, inf])
    w = np.float32([1.0, 1.0, 0.5, -0.5, 0.3, 0.0, 1 + 2 ** -20, inf, -inf, inf, inf])
    sa = _split3_planes(a)
    sw = _split3_planes(w, weight_side=True)
    kept = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]
    with np.errstate(invalid="ignore"):
        got = sum(sa[i].astype(np.float64) * sw[j].astype(np.float64) for i, j in kept)
        want = a.astype(np.float64) * w.astype(np.float64)
    row_side = np.isfinite(w)          # first seven cases: the non-finite value is on the row side
    assert np.array_equal(np.isnan(got[row_side]), np.isnan(want[row_side]))
    assert np.array_equal(got[row_side][~np.isnan(want[row_side])], want[row_side][~np.isnan(want[row_side])])
    assert np.all(np.isnan(got[~row_side]))            # weight side: NaN (fp32: +-inf or NaN) — never finite
    assert not np.any(np.isfinite(want[~row_side]))
    # a NaN whose payload sits in the low 16 bits must stay a NaN after truncation to bf16
    sneaky = np.uint32([0x7F800001]).view(np.float32)
    assert np.isnan(_split3_planes(sneaky)[2][0])
