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


def test_split_image_layout_of_the_build_kernel():
    """The build kernel's operand (csrc/as_k2bf.hip, split_bf16_kernel): per row and 32-column slab 32 bf16 heads then 32 bf16
    tails -- the 128 bytes of the fp32 slab row; 16-byte chunk c holds heads (c < 4) or tails (c >= 4) of columns
    8 (c & 3) .. 8 (c & 3) + 7, one lane's operand of a 16-column k-step of v_mfma_f32_32x32x16_bf16 (lane half h of k-step s
    reads chunk 2 s + h and 4 + 2 s + h).  Restated here: image -> (head, tail) recovers the split of every element."""
    rng = np.random.default_rng(3)
    rows, dp = 5, 96
    x = rng.standard_normal((rows, dp)).astype(np.float32)
    hi, lo = split(x)
    img = np.zeros((rows, dp * 2), dtype=np.uint16)           # 4 bytes per element, as bf16 halves
    for r in range(rows):
        for s in range(dp // 32):
            img[r, 64 * s: 64 * s + 32] = (hi[r, 32 * s: 32 * s + 32].view(np.uint32) >> 16).astype(np.uint16)
            img[r, 64 * s + 32: 64 * s + 64] = (lo[r, 32 * s: 32 * s + 32].view(np.uint32) >> 16).astype(np.uint16)
    for r in range(rows):
        for s in range(dp // 32):
            for kstep in range(2):
                for h in range(2):
                    c_head, c_tail = 2 * kstep + h, 4 + 2 * kstep + h
                    cols = 32 * s + 16 * kstep + 8 * h + np.arange(8)
                    got_h = (img[r, 64 * s + 8 * c_head: 64 * s + 8 * c_head + 8].astype(np.uint32) << 16).view(np.float32)
                    got_l = (img[r, 64 * s + 8 * c_tail: 64 * s + 8 * c_tail + 8].astype(np.uint32) << 16).view(np.float32)
                    np.testing.assert_array_equal(got_h, hi[r, cols])
                    np.testing.assert_array_equal(got_l, lo[r, cols])


def test_build_kernel_error_coefficient_covers_the_three_product_key():
    """err_coef of the bf16 build kernel (csrc/as_build.hip): |key32 - key64| <= [(6 Dp + 32) 2^-24 + 3.03 2^-16] (n_i + n_j)
    for key = n_i + n_j - 2 G with G the fp32-accumulated sum of xh.yh + xh.yl + xl.yh.  Restated with sequential fp32
    accumulation in the kernel's order (per 16-column k-step: the three products of its columns), on clustered, scaled,
    near-duplicate and cancelling rows."""
    rng = np.random.default_rng(5)
    worst = 0.0
    for trial in range(60):
        d = int(rng.choice([32, 96, 768]))
        x = rng.standard_normal(d)
        y = rng.standard_normal(d)
        kind = trial % 4
        if kind == 1:
            y = x + 1e-3 * y                      # near-duplicates: the key cancels to ~1e-6 of the norms
        elif kind == 2:
            x *= 1e3
            y *= 1e-3
        elif kind == 3:
            y = -x + 1e-2 * y
        x32, y32 = x.astype(np.float32), y.astype(np.float32)
        xh, xl = split(x32)
        yh, yl = split(y32)
        acc = np.float32(0.0)
        for s in range(0, d, 16):
            for a, b in ((xh, yh), (xh, yl), (xl, yh)):
                for c in range(s, min(s + 16, d)):
                    acc = np.float32(acc + np.float32(a[c]) * np.float32(b[c]))     # bf16 x bf16 is exact in fp32; one rounding per addition
        ni, nj = np.float32(np.sum(x32.astype(np.float64) ** 2)), np.float32(np.sum(y32.astype(np.float64) ** 2))
        key32 = np.float32(np.float32(ni + nj) + np.float32(-2.0) * acc)
        key64 = float(np.sum((x32.astype(np.float64) - y32.astype(np.float64)) ** 2))
        dp = (d + 31) // 32 * 32
        coef = (6 * dp + 32) * 2.0 ** -24 + 3.03 * 2.0 ** -16
        bound = coef * (float(ni) + float(nj))
        assert abs(float(key32) - key64) <= bound, (trial, kind, abs(float(key32) - key64) / (float(ni) + float(nj)), coef)
        worst = max(worst, abs(float(key32) - key64) / (float(ni) + float(nj)) / coef)
    assert worst < 1.0
