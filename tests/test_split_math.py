"""The arithmetic claim behind csrc/gemm_split.hip (precision='split'), checked in numpy on the CPU: an fp32 value is
EXACTLY the sum of three bf16 terms obtained by round-to-nearest of successive remainders (for |a| in [1e-33, 3.38e38]:
below, the third term falls into the bf16 subnormals; above, the first rounds to infinity), and the six term products the
kernel accumulates miss the exact product by less than one fp32 rounding."""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest-even bfloat16, returned as float32 (what v_cvt_pk_bf16_f32 does for finite values)."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7fff + ((u >> 16) & 1)) >> 16 << 16
    return u.astype(np.uint32).view(np.float32).reshape(np.shape(x))


def split3(a):
    h = bf16_rne(a)
    r = (a - h).astype(np.float32)          # exact: fp32 subtraction of a value from its own rounding
    m = bf16_rne(r)
    l = (r - m).astype(np.float32)
    return h, m, l


def _samples(n, seed):
    rng = np.random.RandomState(seed)
    a = rng.standard_normal(n).astype(np.float32) * np.exp(rng.uniform(-30, 30, n)).astype(np.float32)
    edge = np.array([1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0e38, 1.0e-30, 0.0, 255.5, 0.1, 16777215.0], np.float32)
    return np.concatenate([a, edge])


def test_three_bf16_terms_reproduce_fp32_exactly():
    a = _samples(400000, 1)
    h, m, l = split3(a)
    for t in (h, m, l):                                        # every term IS a bf16 value
        assert np.array_equal(bf16_rne(t), t)
    s = h.astype(np.float64) + m.astype(np.float64) + l.astype(np.float64)
    assert np.array_equal(s, a.astype(np.float64))             # no bit lost
    nz = a != 0
    assert (np.abs(m[nz]) <= 2.0 ** -8 * np.abs(a[nz]) * (1 + 2.0 ** -7)).all()     # |a1| <= 2^-9 |a0| ~ 2^-8.99 |a|
    assert (np.abs(l[nz]) <= 2.0 ** -16 * np.abs(a[nz])).all()


def test_six_products_are_within_one_fp32_rounding_of_the_product():
    a, b = _samples(200000, 2), _samples(200000, 3)[::-1].copy()
    keep = (np.abs(a) > 1e-18) & (np.abs(a) < 1e18) & (np.abs(b) > 1e-18) & (np.abs(b) < 1e18)   # products stay normal
    a, b = a[keep], b[keep]
    ah, am, al = (t.astype(np.float64) for t in split3(a))
    bh, bm, bl = (t.astype(np.float64) for t in split3(b))
    six = ah * bh + ah * bm + am * bh + am * bm + ah * bl + al * bh          # each product exact in fp32 (8 x 8 bits)
    exact = a.astype(np.float64) * b.astype(np.float64)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() < 2.0 ** -24                                            # below half an ulp of the fp32 product
    # and every single term product fits fp32 exactly
    for x, y in ((ah, bh), (ah, bm), (am, bm), (al, bh)):
        p = x * y
        assert np.array_equal(p.astype(np.float32).astype(np.float64), p)


def test_dot_product_error_matches_fp32():
    """A 4096-long dot product: the six-product sum accumulated in fp32 is as close to float64 as the plain fp32 dot."""
    rng = np.random.RandomState(4)
    A = rng.standard_normal((64, 4096)).astype(np.float32)
    B = rng.standard_normal((4096, 32)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    mag = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    Ah, Am, Al = split3(A)
    Bh, Bm, Bl = split3(B)
    acc = np.zeros((64, 32), np.float32)
    for k0 in range(0, 4096, 16):                                           # k-blocks of one MFMA, fp32 accumulator
        sl = slice(k0, k0 + 16)
        for X, Y in ((Al, Bh), (Ah, Bl), (Am, Bm), (Am, Bh), (Ah, Bm), (Ah, Bh)):
            acc = (acc + (X[:, sl].astype(np.float64) @ Y[sl].astype(np.float64)).astype(np.float32)).astype(np.float32)
    plain = np.zeros((64, 32), np.float32)
    for k0 in range(0, 4096, 16):
        sl = slice(k0, k0 + 16)
        plain = (plain + (A[:, sl].astype(np.float64) @ B[sl].astype(np.float64)).astype(np.float32)).astype(np.float32)
    e_split = (np.abs(acc - ref) / mag).max()
    e_plain = (np.abs(plain - ref) / mag).max()
    assert e_split < 2.0 ** -21 and e_split <= 3.0 * e_plain + 2.0 ** -26, (e_split, e_plain)
