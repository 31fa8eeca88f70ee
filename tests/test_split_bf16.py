"""The arithmetic of dd_conv3x3_mfma (csrc/dd_conv_mfma.hip) on the CPU, through its NumPy restatement oracle/ref_split_bf16.py:
three bf16 pieces per fp32 operand, six partial products per multiply, fp32 accumulation -- exact split, fp32-grade dot products."""
import numpy as np

from oracle import ref_split_bf16 as R


def _values(rs, n):
    x = rs.standard_normal(n).astype(np.float32) * np.exp(rs.uniform(-30, 30, n)).astype(np.float32)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -10, 1.0 + 2.0 ** -20, 1.0 + 2.0 ** -23, 3.0e38, -3.0e38, 1.1754944e-38, 255.99998, 0.33333334],
                    dtype=np.float32)
    return np.concatenate([x, edge])


def test_split_is_exact_and_every_piece_is_bf16():
    x = _values(np.random.RandomState(0), 200000)
    x1, x2, x3 = R.split3(x)
    for p in (x1, x2, x3):
        assert np.all((p.view(np.uint32) & 0xFFFF) == 0)                    # representable in bf16
    s = (x1.astype(np.float64) + x2.astype(np.float64) + x3.astype(np.float64))
    assert np.array_equal(s.astype(np.float32), x) and np.array_equal(s, x.astype(np.float64))      # exact, not merely to fp32 rounding
    # the pieces shrink by at least 2^-8 each (round to nearest: half an ulp of the piece above)
    nz = x != 0
    assert np.all(np.abs(x2[nz]) <= np.abs(x1[nz]) * 2.0 ** -8) and np.all(np.abs(x3[nz]) <= np.abs(x1[nz]) * 2.0 ** -16)


def test_bf16_rounding_is_nearest_even():
    # 1 + 2^-8 lies exactly between two bf16 values: ties go to the even significand (1.0); 1 + 3 * 2^-8 goes up to 1 + 2^-6
    assert R.bf16_round(np.float32(1.0 + 2.0 ** -8)) == np.float32(1.0)
    assert R.bf16_round(np.float32(1.0 + 3.0 * 2.0 ** -8)) == np.float32(1.0 + 2.0 ** -6)
    assert R.bf16_round(np.float32(1.0 + 2.0 ** -8 + 2.0 ** -20)) == np.float32(1.0 + 2.0 ** -7)


def test_dropped_cross_terms_are_below_fp32_rounding():
    rs = np.random.RandomState(1)
    a, b = _values(rs, 50000)[:50000], _values(rs, 50000)[:50000]
    pa, pb = R.split3(a), R.split3(b)
    kept = sum(pa[i].astype(np.float64) * pb[j].astype(np.float64) for i, j in R.PRODUCTS)
    exact = a.astype(np.float64) * b.astype(np.float64)
    ok = np.isfinite(exact) & (np.abs(exact) > 1e-30) & (np.abs(exact) < 1e30)
    rel = np.abs(kept - exact)[ok] / np.abs(exact)[ok]
    assert rel.max() < 2.0 ** -24            # half an ulp of fp32: the six products are closer to a*b than fl32(a*b) is guaranteed to be
    print("largest relative size of the three dropped cross terms: 2^%.1f" % np.log2(rel.max()))


def test_dot_product_is_as_accurate_as_an_fp32_chain():
    rs = np.random.RandomState(2)
    for K in (576, 1152, 4608):                  # 9 taps x 64 / 128 / 512 channels
        a = rs.standard_normal((256, K)).astype(np.float32)
        b = (rs.standard_normal((256, K)) / np.sqrt(K)).astype(np.float32)
        ref = (a.astype(np.float64) * b.astype(np.float64)).sum(-1)
        e_split = np.abs(R.dot_split(a, b) - ref).max() / np.abs(ref).max()
        e_fp32 = np.abs(R.dot_fp32(a, b) - ref).max() / np.abs(ref).max()
        print("K=%d  split-bf16 %.2e  fp32 chain %.2e" % (K, e_split, e_fp32))
        assert e_split <= 1.5 * e_fp32 and e_split < 2e-6


def test_small_integers_are_exact():
    rs = np.random.RandomState(3)
    a = rs.randint(-7, 8, (64, 288)).astype(np.float32)
    b = rs.randint(-3, 4, (64, 288)).astype(np.float32)
    assert np.array_equal(R.dot_split(a, b), (a.astype(np.float64) * b.astype(np.float64)).sum(-1).astype(np.float32))
