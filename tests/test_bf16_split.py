"""CPU: the error bound behind the batched pass's products on the bf16 matrix pipe (`split_bf16` /
`coef_query` in csrc/as_scan.hip), restated in numpy.  Every fp32 operand is a bf16 head plus the bf16 of what the head
leaves, v = hi + lo + r with |r| <= 2^-16 |v|; the kernel forms qh.xh + qh.xl + ql.xh (bf16 x bf16 is exact in fp32) and
drops ql.xl and the two remainders: at most 3 * 2^-16 * 1.01 of sum |q_k x_k| -- the representation part of the
coefficient the prefilter's bound and the a-posteriori proof are widened by."""
import numpy as np


def to_bf16(v):
    """Round-to-nearest-even of fp32 values to bf16 (8 significant bits), returned as fp32."""
    u = np.asarray(v, dtype=np.float32).view(np.uint32).astype(np.uint64)
    lsb = (u >> 16) & 1
    r = ((u + 0x7FFF + lsb) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def split(v):
    v = np.asarray(v, dtype=np.float32)
    hi = to_bf16(v)
    rem = v - hi                      # exact in fp32 (the kernel relies on it)
    assert np.array_equal(rem.astype(np.float64), v.astype(np.float64) - hi.astype(np.float64))
    lo = to_bf16(rem)
    return hi, lo


def test_head_plus_tail_leaves_at_most_two_to_the_minus_sixteen():
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.standard_normal(200000), rng.standard_normal(1000) * 1e-20, rng.standard_normal(1000) * 1e20,
                        np.float32([1.0, 1.0 + 2.0 ** -8, 1.0 + 2.0 ** -7 + 2.0 ** -23, 255.0 / 256.0, 3.0, -0.1])]).astype(np.float32)
    hi, lo = split(v)
    r = v.astype(np.float64) - hi.astype(np.float64) - lo.astype(np.float64)
    assert np.all(np.abs(r) <= 2.0 ** -16 * np.abs(v.astype(np.float64)))
    assert np.all(np.abs(lo.astype(np.float64)) <= 2.0 ** -8 * (1 + 2.0 ** -8) * np.abs(v.astype(np.float64)))


def test_three_products_are_within_the_coefficient_of_the_true_dot():
    rng = np.random.default_rng(1)
    worst = 0.0
    for trial in range(200):
        d = int(rng.choice([32, 384, 768]))
        kind = trial % 4
        q = rng.standard_normal(d)
        x = rng.standard_normal(d)
        if kind == 1:                       # one dominant component each
            q[3] += 50.0
            x[3] += 50.0
        elif kind == 2:                     # values just under a bf16 rounding boundary: the largest relative remainders
            q = np.sign(q) * (1.0 + 2.0 ** -8 - 2.0 ** -20) * 2.0 ** rng.integers(-6, 6, d)
            x = np.sign(x) * (1.0 + 2.0 ** -8 - 2.0 ** -20) * 2.0 ** rng.integers(-6, 6, d)
        elif kind == 3:                     # cancelling dot: the bound is relative to sum |q_k x_k|, not to the dot
            x = np.concatenate([q[: d // 2], -q[: d // 2]]) + 1e-3 * x
        q32, x32 = q.astype(np.float32), x.astype(np.float32)
        qh, ql = split(q32)
        xh, xl = split(x32)
        f = lambda a: a.astype(np.float64)
        kept = np.sum(f(qh) * f(xh)) + np.sum(f(qh) * f(xl)) + np.sum(f(ql) * f(xh))      # exact products, exact sum: representation only
        true = np.sum(f(q32) * f(x32))
        scale = np.sum(np.abs(f(q32) * f(x32)))
        assert abs(kept - true) <= 3.0 * 2.0 ** -16 * 1.01 * scale, (trial, kind, abs(kept - true) / scale)
        worst = max(worst, abs(kept - true) / scale)
        assert scale <= np.linalg.norm(f(q32)) * np.linalg.norm(f(x32)) * (1 + 1e-12)     # Cauchy-Schwarz: what coef_query multiplies
    assert worst > 2.0 ** -19        # (the bound is not vacuous: the adversarial operands come within a factor of it)


def test_non_finite_heads_keep_a_zero_tail():
    """inf - inf would be NaN: the kernel gives a non-finite value a zero tail (the head carries it)."""
    v = np.float32([np.inf, -np.inf, np.nan, 1.5])
    hi = to_bf16(np.where(np.isfinite(v), v, 0)).copy()
    hi[~np.isfinite(v)] = v[~np.isfinite(v)]
    with np.errstate(invalid="ignore"):
        rem = v - hi
    lo = np.where(rem == rem, rem, np.float32(0))
    assert lo[0] == 0 and lo[1] == 0 and lo[2] == 0 and lo[3] == 0
