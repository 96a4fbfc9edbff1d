"""The arithmetic claim behind the certified far planes of the default hot loop (csrc/rt_fastdiv.hpp), checked on the CPU in IEEE float32
(numpy): with r = RN(1/d), the product t' = RN(n * r) stays within 3.02 u of the quotient RN(n / d) (u = 2^-24), minima of three inherit the bound, and whenever
the kernel's certificate |min3 t' - tmin| > 2^-21 * max|min3 t'| holds, `tmin <= min3 t'` IS `tmin <= min3 RN(n/d)` — for random operands of the fast-division
class and for tmin placed adversarially within a few ulp of the far parameters.  The signs of product and quotient agree and they are zero together."""
import numpy as np

U = np.float32(2.0 ** -24)
EPS = np.float32(2.0 ** -21)   # RT_FAR_EPS


def _class_operands(rng, n):
    """plane offsets n (0 or 2^-64 <= |n| < 2^41) and directions d (2^-40 <= |d| < 2^40) of the fast-division class"""
    def mag(lo, hi, size):
        e = rng.uniform(lo, hi, size)
        return (np.exp2(e) * rng.uniform(1.0, 2.0, size)).astype(np.float32)
    sgn = lambda size: np.where(rng.random(size) < 0.5, -1.0, 1.0).astype(np.float32)
    nn = mag(-20, 12, n) * sgn(n)
    wide = rng.random(n) < 0.1
    nn[wide] = (mag(-64, 40, wide.sum()) * sgn(wide.sum()))
    nn[rng.random(n) < 0.01] = 0.0
    d = mag(-8, 8, n) * sgn(n)
    widd = rng.random(n) < 0.1
    d[widd] = mag(-40, 39, widd.sum()) * sgn(widd.sum())
    return nn, d


def test_product_stays_within_three_roundings_of_the_quotient():
    rng = np.random.default_rng(5)
    n, d = _class_operands(rng, 1 << 22)
    r = np.float32(1.0) / d
    t, q = n * r, n / d
    assert np.all(np.isfinite(t)) and np.all(np.isfinite(q))
    assert np.array_equal(np.sign(t), np.sign(q)) and np.array_equal(t == 0, q == 0)     # sign-exact, zero together
    err = np.abs(t.astype(np.float64) - q.astype(np.float64))
    assert np.all(err <= 3.02 * float(U) * np.abs(n.astype(np.float64) / d.astype(np.float64)) + 0.0)


def test_certificate_implies_the_exact_decision():
    rng = np.random.default_rng(6)
    m = 1 << 21
    n, d = _class_operands(rng, 3 * m)
    n, d = n.reshape(m, 3), d.reshape(m, 3)
    r = np.float32(1.0) / d
    far_p, far_q = (n * r).min(axis=1), (n / d).min(axis=1)
    assert np.all(np.abs(far_p.astype(np.float64) - far_q) <= 3.1 * float(U) * np.abs(far_p.astype(np.float64)))
    # tmin: random, and (half of the cases) within a few ulp of the far parameter — where the certificate must refuse
    tmin = (far_q * rng.uniform(-2, 3, m)).astype(np.float32)
    k = rng.integers(-6, 7, m // 2)
    tmin[: m // 2] = np.nextafter(far_q[: m // 2], np.where(k > 0, np.float32(np.inf), np.float32(-np.inf)).astype(np.float32))
    for _ in range(5):   # walk up to 6 ulp away
        step = np.abs(k) > 1
        tmin[: m // 2][step] = np.nextafter(tmin[: m // 2][step], np.where(k[step] > 0, np.float32(np.inf), np.float32(-np.inf)).astype(np.float32))
        k = k - np.sign(k) * step
    # the kernel scales with the larger of the two boxes' far parameters: any scale >= |far_p| keeps the claim, the test uses the tightest one
    gap = np.abs(far_p - tmin)
    certain = gap > np.abs(far_p) * EPS
    assert 0.3 < certain.mean() < 0.9
    assert np.array_equal((tmin <= far_p)[certain], (tmin <= far_q)[certain])
    # and the adversarial half really contains disagreements that the certificate refused
    disagree = (tmin <= far_p) != (tmin <= far_q)
    assert disagree.any() and not (disagree & certain).any()
