"""GPU: edge cases of the hot path against the oracle -- tile boundaries, odd feature
counts, tiny inputs, isolated graphs, exact duplicates (ties), wide rows (generic scan)."""
import numpy as np
import pytest

from conftest import assert_hits_match, calibrate_eps, clustered

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def _compare(X, gp, oracle_lib, queries, taus=(1.0, 0.62, 0.0)):
    import pyarrowspace_amd as asp
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    ref = oracle_lib.OracleIndex(X, gp)
    np.testing.assert_allclose(aspace.lambdas(), ref.lambdas, rtol=RTOL, atol=1e-300)
    np.testing.assert_allclose(gl.degrees(), ref.deg, rtol=RTOL, atol=1e-300)
    indptr, indices, values = gl.to_csr()
    rows = np.repeat(np.arange(X.shape[0]), np.diff(indptr))
    assert np.array_equal(indices[indices != rows], ref.indices)
    for q in queries:
        q = np.ascontiguousarray(q, dtype=np.float64)
        for tau in taus:
            try:
                want, lq = ref.search(q, tau)
            except oracle_lib.ZeroLambda:
                with pytest.raises(asp.PanicException):
                    aspace.search(q, gl, tau)
                continue
            got = aspace.search(q, gl, tau)
            assert_hits_match(got, want, ref.scores(q, tau, lq), rtol=RTOL)
    return aspace, gl, ref


@pytest.mark.parametrize("n", [2, 3, 63, 64, 65, 255, 256, 257, 511, 513])
def test_row_tile_boundaries(oracle_lib, n):
    d = 40
    X = clustered(n, d, nclust=3, seed=n)
    k = min(6, n)
    gp = {"eps": calibrate_eps(X, max(1, min(k, n - 1))), "k": k, "topk": 4, "p": 2.0, "sigma": None}
    _compare(X, gp, oracle_lib, [X[0] * 1.01, X[n - 1] + 0.01])


@pytest.mark.parametrize("d", [1, 2, 3, 5, 31, 32, 33, 255, 257, 1000])
def test_feature_counts(oracle_lib, d):
    n = 300
    X = clustered(n, d, nclust=4, seed=100 + d, normalise=d > 1)
    if d == 1:
        X = X + 3.0
    gp = {"eps": calibrate_eps(X, 5), "k": 5, "topk": 5, "p": 2.0, "sigma": None}
    _compare(X, gp, oracle_lib, [X[7] * 1.02, X[100] + 0.01 / np.sqrt(d)])


def test_wide_rows_use_the_generic_scan(oracle_lib):
    n, d = 200, 2100                      # dp = 2112 floats > 8 x 256: generic scan kernel
    X = clustered(n, d, nclust=4, seed=77)
    gp = {"eps": calibrate_eps(X, 4), "k": 4, "topk": 3, "p": 2.0, "sigma": None}
    _compare(X, gp, oracle_lib, [X[3] * 1.01])


def test_single_item_and_no_edges(oracle_lib):
    import pyarrowspace_amd as asp
    X = np.array([[1.0, 2.0, 3.0]])
    aspace, gl = asp.ArrowSpaceBuilder.build({"eps": 1.0, "k": 3, "topk": 2, "p": 2.0}, X)
    assert aspace.nitems == 1 and gl.nnodes == 1 and aspace.lambdas().tolist() == [0.0]
    hits = aspace.search(np.array([1.0, 2.0, 3.1]), gl, 1.0)   # the item is within eps of the query
    assert len(hits) == 1 and hits[0][0] == 0
    # eps so small that no pair is connected: every lambda is 0, tau0 floors, queries hit the zero-lambda assert
    Y = clustered(100, 8, nclust=2, seed=1)
    a2, g2 = asp.ArrowSpaceBuilder.build({"eps": 1e-9, "k": 3, "topk": 2, "p": 2.0}, Y)
    assert not a2.lambdas().any() and not g2.degrees().any() and g2.tau0 == 1e-12
    with pytest.raises(asp.PanicException):
        a2.search(np.ascontiguousarray(Y[0] * 1.5), g2, 0.5)


def test_exact_duplicates_break_ties_by_index(oracle_lib):
    rng = np.random.default_rng(0)
    base = rng.standard_normal((6, 16))
    X = np.repeat(base, 50, axis=0)       # 300 rows, 6 distinct points x 50 copies: massive exact ties
    gp = {"eps": 0.5, "k": 5, "topk": 6, "p": 2.0, "sigma": 0.3}
    _compare(X, gp, oracle_lib, [base[2] + 0.01, base[5]])


@pytest.mark.parametrize("metric", ["l2", "cosine"])
def test_duplicate_heavy_build_is_settled_by_the_band_pass(oracle_lib, metric):
    """5 200 items, 3 000 of them exact copies in groups of 30, plus 200 near-copies that fp32 cannot tell apart: the
    first k-NN pass cannot prove those rows exact (ties at the k-th distance); the second pass collects every
    column inside the proven band and settles them on the device -- no row is left to the row-serial fp64 path."""
    import pyarrowspace_amd as asp
    rng = np.random.default_rng(8)
    n, d, k = 5200, 96, 10
    X = clustered(n, d, nclust=20, seed=8)
    src = rng.choice(2000, 100, replace=False)
    for g, s0 in enumerate(src):
        X[2000 + 30 * g: 2000 + 30 * (g + 1)] = X[s0]
    # near-copies fp32 cannot tell apart and fp64 can (1 - cos of vectors 1e-9 apart is below fp64's own resolution:
    # the cosine case keeps them 1e-4 apart, 1 - cos ~ 5e-9 against an fp32 key error of 7e-6)
    amp = 1e-9 if metric == "l2" else 1e-4
    X[2000 + 3000: 2000 + 3000 + 200] = X[src[0]] + amp * rng.standard_normal((200, d)) / np.sqrt(d)
    # eps reaches the copies and near-copies only (a quantile-calibrated eps is 0 here); everything else is isolated
    gp = {"eps": 1e-6, "k": k, "topk": 8, "p": 2.0, "sigma": None, "metric": metric}
    aspace, gl, ref = _compare(X, gp, oracle_lib, [X[src[3]], X[5100]], taus=(1.0, 0.62))
    st = gl.build_stats()
    assert st["fallback_rows"] == 0 and st["fallback_s"] == 0.0, st


def test_unnormalised_scaled_items(oracle_lib):
    """The reference's harnesses feed x100-scaled, unnormalised embeddings (tests/test_3_beir.py:155-156,190)."""
    n, d = 500, 96
    X = clustered(n, d, nclust=6, seed=3, normalise=False) * 100.0
    gp = {"eps": calibrate_eps(X, 8), "k": 8, "topk": 5, "p": 2.0, "sigma": None}
    _compare(X, gp, oracle_lib, [X[11] * 1.001, X[400] + 0.5])


def test_p_not_two_and_explicit_sigma(oracle_lib):
    n, d = 400, 32
    X = clustered(n, d, nclust=5, seed=8)
    eps = calibrate_eps(X, 6)
    for kern in ("gaussian", "rational"):
        gp = {"eps": eps, "k": 6, "topk": 4, "p": 3.0, "sigma": eps * 0.8, "kernel": kern}
        _compare(X, gp, oracle_lib, [X[9] * 1.01], taus=(0.62,))


@pytest.mark.parametrize("mode", [0, 1])
def test_dense_neighbourhoods_prune_thousands_of_candidates(oracle_lib, mode):
    """eps admits ~2000 candidates per query: the finish kernels prune by radix select before ranking."""
    n, d = 6000, 48
    X = clustered(n, d, nclust=2, noise=0.3, seed=19)
    gp = {"eps": 0.9, "k": 12, "topk": 9, "p": 2.0, "sigma": None, "_search_mode": mode}
    _compare(X, gp, oracle_lib, [X[5] * 1.01, X[n - 3] + 0.01 / np.sqrt(d)], taus=(0.62, 1.0))


def test_every_pair_within_eps(oracle_lib):
    """eps above the diameter (the reference harnesses' eps=10 on small-norm data, tests/test_3_beir.py:194-200):
    the k cap alone decides the graph, and every query overflows the prefilter buffer into the robust path."""
    n, d = 5000, 64
    X = clustered(n, d, nclust=6, noise=0.4, seed=23)
    gp = {"eps": 10.0, "k": 25, "topk": 15, "p": 2.0, "sigma": None}
    _compare(X, gp, oracle_lib, [X[17] * 1.02, X[n - 1] + 0.02 / np.sqrt(d)], taus=(0.62, 1.0))


@pytest.mark.parametrize("topk", [57, 100, 1000, 1024])
def test_large_topk(oracle_lib, topk):
    """topk up to 1024 (candidate list = topk + margin, exact re-scoring in rounds of 64)."""
    n, d = 5000, 64
    X = clustered(n, d, nclust=10, seed=23)
    gp = {"eps": calibrate_eps(X, 8), "k": 8, "topk": topk, "p": 2.0, "sigma": None}
    aspace, gl, ref = _compare(X, gp, oracle_lib, [X[17] * 1.01, X[4000] + 0.01 / np.sqrt(d)])
    got = aspace.search_batch(np.stack([X[17] * 1.01, X[3] * 0.99, X[99]]), gl, 0.62)
    for b, q in enumerate([X[17] * 1.01, X[3] * 0.99, X[99]]):
        want, lq = ref.search(np.ascontiguousarray(q), 0.62)
        assert_hits_match(got[b], want, ref.scores(np.ascontiguousarray(q), 0.62, lq), rtol=RTOL)


def test_limits_are_reported_as_value_errors():
    import pyarrowspace_amd as asp
    X = clustered(300, 16, nclust=3, seed=1)
    a, g = asp.ArrowSpaceBuilder.build({"eps": 0.8, "k": 5, "topk": 2000, "p": 2.0}, X)   # topk > nitems is clamped
    assert len(a.search(np.ascontiguousarray(X[0]), g, 1.0)) == 300
    with pytest.raises(ValueError, match="k"):
        asp.ArrowSpaceBuilder.build({"eps": 0.8, "k": 130, "topk": 5, "p": 2.0}, X)
    Y = clustered(1500, 16, nclust=3, seed=1)
    with pytest.raises(ValueError, match="topk"):   # refused before any upload or GPU work, not at the first search
        asp.ArrowSpaceBuilder.build({"eps": 0.8, "k": 5, "topk": 1300, "p": 2.0}, Y)


@pytest.mark.parametrize("keep64", [False, True])
def test_save_and_load_round_trip(tmp_path, keep64):
    """Extension: a loaded index answers exactly like the one that was saved."""
    import pyarrowspace_amd as asp
    n, d = 2000, 72
    X = clustered(n, d, nclust=6, seed=29)
    if not keep64:
        X = X.astype(np.float32).astype(np.float64)      # exactly fp32-representable: no fp64 copy is kept
    gp = {"eps": calibrate_eps(X, 7), "k": 7, "topk": 9, "p": 2.0, "sigma": None, "metric": "cosine" if keep64 else "l2"}
    a, g = asp.ArrowSpaceBuilder.build(gp, X)
    path = str(tmp_path / "index.asidx")
    a.save(g, path)
    b, h = asp.ArrowSpaceBuilder.load(path)
    assert (b.nitems, b.nfeatures, h.nnodes, h.graph_params, h.tau0) == (a.nitems, a.nfeatures, g.nnodes, g.graph_params, g.tau0)
    np.testing.assert_array_equal(b.lambdas(), a.lambdas())
    for x, y in zip(h.to_csr(), g.to_csr()):
        np.testing.assert_array_equal(x, y)
    np.testing.assert_array_equal(b.get_item(5)[0], X[5])
    rng = np.random.default_rng(1)
    for _ in range(4):
        q = np.ascontiguousarray(X[rng.integers(0, n)] * 1.01 + 0.002 * rng.standard_normal(d))
        for tau in (1.0, 0.62):
            assert b.search(q, h, tau) == a.search(q, g, tau)
    with pytest.raises(ValueError):
        asp.ArrowSpaceBuilder.load(str(tmp_path / "missing.asidx"))


@pytest.mark.parametrize("n,d", [(140037, 100), (135001, 300), (133333, 700), (132000, 1000)])
def test_scan_rounds_and_remainder(oracle_lib, n, d):
    """Sizes just above one full round of the LDS-DMA scan (2048 waves x 64 rows = 131072): every wave takes one
    round-robin chunk plus its share of the remainder; rows of 1..4 KiB pick the four ring shapes.  The all-pairs
    oracle build is out of reach at this size, so the CPU scorer runs over the GPU-built lambdas and degrees
    (the graph itself is covered by the smaller cases) and checks the search end to end."""
    import torch
    import bench
    import pyarrowspace_amd as asp
    dev = torch.device("cuda", 0)
    X = bench.make_data(n, d, 7, dev, nclust=256)
    gp = {"eps": bench.calibrate_eps(X, 10), "k": 10, "topk": 8, "p": 2.0, "sigma": None}
    aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, d, d)
    Xh = X.double().cpu().numpy()
    ref = oracle_lib.OracleSearchOnly(Xh, gp, gl.degrees(), aspace.lambdas(), gl.tau0)
    rng = np.random.default_rng(3)
    rows = [0, n - 1, 131071, 131072, int(rng.integers(0, n)), int(rng.integers(0, n))]
    Q = np.stack([Xh[i] * 1.01 + 0.01 * rng.standard_normal(d) / np.sqrt(d) for i in rows])
    for q in Q:
        for tau in (0.62, 1.0):
            want, lq = ref.search(q, tau)
            got = aspace.search(np.ascontiguousarray(q), gl, tau)
            assert_hits_match(got, want, ref.scores(q, tau, lq), rtol=RTOL)
    if d <= 1024:
        got = aspace.search_batch(Q, gl, 0.62)
        for b, q in enumerate(Q):
            want, lq = ref.search(q, 0.62)
            assert_hits_match(got[b], want, ref.scores(q, 0.62, lq), rtol=RTOL)


def test_threads_share_a_space(oracle_lib):
    """ctypes drops the GIL inside as_search: the library serialises per space, results must not cross threads
    (the reference holds the GIL throughout, src/lib.rs:132-174)."""
    import threading
    import pyarrowspace_amd as asp
    n, d = 4000, 96
    X = clustered(n, d, nclust=8, seed=1)
    gp = {"eps": calibrate_eps(X, 8), "k": 8, "topk": 6, "p": 2.0, "sigma": None}
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    ref = oracle_lib.OracleIndex(X, gp)
    want = {i: ref.search(X[i] * 1.01, 0.62)[0] for i in range(16)}
    errs = []

    def worker(t):
        try:
            for s in range(50):
                i = (t * 7 + s) % 16
                got = aspace.search(X[i] * 1.01, gl, 0.62)
                if [j for j, _ in got] != [j for j, _ in want[i]]:
                    errs.append((t, i, got))
        except BaseException as e:   # noqa: BLE001
            errs.append((t, repr(e)))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs[:2]


def test_two_indexes_alive(oracle_lib):
    import pyarrowspace_amd as asp
    X1 = clustered(3000, 64, nclust=5, seed=2)
    X2 = clustered(2000, 48, nclust=4, seed=3)
    gp1 = {"eps": calibrate_eps(X1, 5), "k": 5, "topk": 4, "p": 2.0, "sigma": None}
    gp2 = {"eps": calibrate_eps(X2, 7), "k": 7, "topk": 5, "p": 2.0, "sigma": None, "metric": "cosine"}
    a1, g1 = asp.ArrowSpaceBuilder.build(gp1, X1)
    a2, g2 = asp.ArrowSpaceBuilder.build(gp2, X2)
    r1, r2 = oracle_lib.OracleIndex(X1, gp1), oracle_lib.OracleIndex(X2, gp2)
    for i in range(4):
        assert [j for j, _ in a1.search(X1[i] * 1.01, g1, 0.62)] == [j for j, _ in r1.search(X1[i] * 1.01, 0.62)[0]]
        assert [j for j, _ in a2.search(X2[i] * 1.01, g2, 0.62)] == [j for j, _ in r2.search(X2[i] * 1.01, 0.62)[0]]
    with pytest.raises(ValueError):
        a1.search(X1[0], g2, 0.62)           # a graph of another space
    del a2, g2
    assert [j for j, _ in a1.search(X1[9] * 1.01, g1, 0.62)] == [j for j, _ in r1.search(X1[9] * 1.01, 0.62)[0]]


def test_non_finite_values_behave_like_the_oracle(oracle_lib):
    """NaN/Inf are counted, not rejected (src/helpers.rs:36-43): a poisoned row drops out of every neighbourhood,
    a poisoned or all-zero query has no neighbours -> the zero-lambda panic."""
    import pyarrowspace_amd as asp
    n, d = 3000, 64
    X = clustered(n, d, nclust=6, seed=4)
    gp = {"eps": calibrate_eps(X, 8), "k": 8, "topk": 5, "p": 2.0, "sigma": None}
    for bad, row in ((np.nan, 11), (np.inf, 12)):
        Xb = X.copy()
        Xb[row, 5] = bad
        aspace, gl = asp.ArrowSpaceBuilder.build(gp, Xb)
        ref = oracle_lib.OracleIndex(Xb, gp)
        np.testing.assert_allclose(aspace.lambdas(), ref.lambdas, rtol=RTOL, atol=1e-300)
        q = np.ascontiguousarray(X[3] * 1.01)
        want, lq = ref.search(q, 0.62)
        assert_hits_match(aspace.search(q, gl, 0.62), want, np.nan_to_num(ref.scores(q, 0.62, lq), nan=-np.inf, posinf=-np.inf), rtol=RTOL)
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    for q in (np.zeros(d), np.r_[np.nan, X[3, 1:]], np.r_[np.inf, X[3, 1:]]):
        with pytest.raises(asp.PanicException):
            aspace.search(np.ascontiguousarray(q), gl, 0.62)


@pytest.mark.parametrize("scale", [1e-22, 1e-6, 1e6, 1e18, 1e20, 1e25])
def test_item_magnitudes_outside_the_fp32_range(oracle_lib, scale):
    """The reference computes in f64 (src/lib.rs:28); squared norms of items scaled by 1e20 overflow fp32 and by
    1e-22 flush to zero, so such an index runs in fp64 end to end instead of trusting fp32 prefilters."""
    n, d = 1500, 48
    X = clustered(n, d, nclust=5, seed=8)
    eps = calibrate_eps(X, 6)
    gp = {"eps": eps * scale, "k": 6, "topk": 4, "p": 2.0, "sigma": None}
    _compare(X * scale, gp, oracle_lib, [X[3] * scale * 1.01, X[n - 2] * scale * 0.99], taus=(0.62, 1.0))


def test_list_wider_than_the_candidate_set(oracle_lib):
    """topk = 1024 over 700 items: every row is a candidate and the list (768 wide) is wider than the set, which is
    itself above the 512-candidate limit where ranking switches to a radix prune -- there is no 768th smallest of 700
    (found by tools/fuzz_parity.py: the prune returned an arbitrary threshold and the re-scoring read wild rows)."""
    n, d = 700, 64
    X = np.random.default_rng(12).standard_normal((n, d))
    gp = {"eps": 12.8, "k": 20, "topk": 1024, "p": 0.5, "sigma": 4.6, "kernel": "rational"}
    aspace, gl, ref = _compare(X, gp, oracle_lib, [X[5] * 1.01, np.random.default_rng(13).standard_normal(d)], taus=(0.62, 1.0))
    assert len(aspace.search(X[5] * 1.01, gl, 0.62)) == n


def test_power_of_two_scaling_changes_nothing():
    """Scaling the items by 2^e (eps and sigma alike) scales every intermediate exactly, in fp32 and fp64: graph,
    degrees and lambdas must come out bit-identical, and a scaled query must return the same hits."""
    import pyarrowspace_amd as asp
    n, d = 2500, 48
    X = clustered(n, d, nclust=6, seed=21)
    eps = calibrate_eps(X, 7)
    gp = {"eps": eps, "k": 7, "topk": 5, "p": 2.0, "sigma": eps * 0.7}
    a0, g0 = asp.ArrowSpaceBuilder.build(gp, X)
    q = np.ascontiguousarray(X[5] * 1.01)
    h0 = a0.search(q, g0, 0.62)
    for e in (-9, 7, 20):
        s = 2.0 ** e
        a1, g1 = asp.ArrowSpaceBuilder.build(dict(gp, eps=eps * s, sigma=eps * 0.7 * s), X * s)
        assert np.array_equal(a1.lambdas(), a0.lambdas()) and np.array_equal(g1.degrees(), g0.degrees())
        assert np.array_equal(g1.to_csr()[1], g0.to_csr()[1]) and np.array_equal(g1.to_csr()[2], g0.to_csr()[2])
        assert a1.search(np.ascontiguousarray(q * s), g1, 0.62) == h0


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_build_from_device_with_a_leading_dimension(oracle_lib, dtype):
    """Items resident in HBM as a view of a wider matrix (ld > d), fp32 or fp64."""
    import torch
    import pyarrowspace_amd as asp
    n, d, ld = 1800, 40, 56
    X = clustered(n, d, nclust=5, seed=33)
    if dtype == "float32":
        X = X.astype(np.float32).astype(np.float64)
    wide = torch.full((n, ld), float("nan"), dtype=getattr(torch, dtype), device="cuda")   # the padding must never be read
    wide[:, :d] = torch.from_numpy(X).to(wide.dtype).cuda()
    gp = {"eps": calibrate_eps(X, 6), "k": 6, "topk": 4, "p": 2.0, "sigma": None}
    aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, wide.data_ptr(), dtype, n, d, ld)
    ref = oracle_lib.OracleIndex(X, gp)
    np.testing.assert_allclose(aspace.lambdas(), ref.lambdas, rtol=RTOL, atol=1e-300)
    v, lam = aspace.get_item(17)
    assert np.array_equal(v, X[17]) and lam == aspace.lambdas()[17]
    q = np.ascontiguousarray(X[9] * 1.01)
    want, lq = ref.search(q, 0.62)
    assert_hits_match(aspace.search(q, gl, 0.62), want, ref.scores(q, 0.62, lq), rtol=RTOL)


@pytest.mark.parametrize("metric,kernel", [("l2", "gaussian"), ("cosine", "rational")])
def test_batched_pass_bf16_products_and_fp32_products_agree(metric, kernel):
    """The batched pass forms its products on the bf16 matrix pipe (every operand as head + tail) and keeps fp16 cosines
    (DESIGN.md 5.5); ARROWSPACE_BATCH_F32_DOTS=1 is round 2's fp32 form.  Both are proven exact behind their own error
    terms: same hits as each other and as the single-query search -- on normalised items, on items whose rows span six
    orders of magnitude, and on rows with one dominant component (the bf16 tails carry everything else)."""
    import os

    import pyarrowspace_amd as asp
    from conftest import calibrate_eps, clustered
    n, d, k, topk = 12000, 768, 12, 10
    rng = np.random.default_rng(8)
    base = clustered(n, d, nclust=24, seed=6)
    spiky = base.copy()
    spiky[:, 5] += 40.0                                   # one component 1 000 x the others
    spread = base * np.exp(rng.uniform(-7, 7, n))[:, None]  # row norms from 1e-3 to 1e3
    for X in (base, spiky, spread):
        gp = {"eps": calibrate_eps(X, k, metric), "k": k, "topk": topk, "p": 2.0, "sigma": None, "metric": metric, "kernel": kernel}
        Q = None
        got = {}
        for form in ("bf16", "fp32"):
            old = os.environ.pop("ARROWSPACE_BATCH_F32_DOTS", None)
            if form == "fp32":
                os.environ["ARROWSPACE_BATCH_F32_DOTS"] = "1"
            try:
                aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)     # (the workspace reads the switch when it is created)
                if Q is None:   # queries next to items that have neighbours (an isolated item's lambda is 0: the zero-lambda panic)
                    rows = rng.choice(np.flatnonzero(gl.degrees() > 0), 40, replace=False)
                    Q = np.stack([X[i] * 1.01 + 0.01 * np.abs(X[i]).mean() * rng.standard_normal(d) for i in rows])
                res = {}
                for tau in (1.0, 0.62, 0.42):
                    res[tau] = aspace.search_batch(Q, gl, tau)
                    assert res[tau] == [aspace.search(q, gl, tau) for q in Q], (form, tau)
                got[form] = res
            finally:
                os.environ.pop("ARROWSPACE_BATCH_F32_DOTS", None)
                if old is not None:
                    os.environ["ARROWSPACE_BATCH_F32_DOTS"] = old
        assert got["bf16"] == got["fp32"]
        assert all(len(h) == topk for h in got["bf16"][0.62])


def test_host_build_streams_chunks_and_matches_the_device_build(monkeypatch):
    """The reference's own call -- build(graph_params, float64 ndarray in host memory, any strides; src/lib.rs:271-277,
    src/helpers.rs:24-46) -- streams the rows through two pinned chunks (as_build -> ingest_host).  With 1 MB chunks a
    20 000 x 96 input takes 15 chunks: same lambdas, Laplacian and items, bit for bit, as the device-resident build of the same
    values; a column-strided view and a row-strided slice of a wider array give the same index; items that do not round-trip
    through fp32 come back from get_item exactly (the second streaming pass into x64)."""
    import torch

    import pyarrowspace_amd as asp
    from conftest import calibrate_eps, clustered
    monkeypatch.setenv("ARROWSPACE_INGEST_CHUNK_MB", "1")
    n, d = 20000, 96
    X32 = clustered(n, d, nclust=20, seed=41).astype(np.float32)
    X = X32.astype(np.float64)                         # lossless in fp32, as embeddings cast up are
    gp = {"eps": calibrate_eps(X, 8, "l2"), "k": 8, "topk": 5, "p": 2.0, "sigma": None}
    Xd = torch.from_numpy(X32).cuda()
    a0, g0 = asp.ArrowSpaceBuilder.build_from_device(gp, Xd.data_ptr(), "float32", n, d, d)
    a1, g1 = asp.ArrowSpaceBuilder.build(gp, X)
    assert np.array_equal(a1.lambdas(), a0.lambdas()) and g1.tau0 == g0.tau0
    for x, y in zip(g1.to_csr(), g0.to_csr()):
        assert np.array_equal(x, y)
    for i in (0, 10921, 10922, n - 1):                 # (chunk boundaries of 1 MB / (96 * 8 B) = 1365 rows and the ends)
        assert np.array_equal(a1.get_item(i)[0], X[i])
    wide = np.zeros((n, 2 * d + 3))
    wide[:, 1:2 * d:2] = X                             # element strides (2 d + 3, 2)
    a2, g2 = asp.ArrowSpaceBuilder.build(gp, wide[:, 1:2 * d:2])
    assert np.array_equal(a2.lambdas(), a0.lambdas())
    wide2 = np.zeros((n, d + 7))
    wide2[:, 3:3 + d] = X                              # unit column stride, row stride d + 7
    a3, g3 = asp.ArrowSpaceBuilder.build(gp, wide2[:, 3:3 + d])
    assert np.array_equal(a3.lambdas(), a0.lambdas())
    q = np.ascontiguousarray(X[77] * 1.001)
    assert a1.search(q, g1, 0.62) == a0.search(q, g0, 0.62) == a2.search(q, g2, 0.62) == a3.search(q, g3, 0.62)
    # fp64 items that fp32 cannot hold: kept in fp64 by the second streaming pass
    Y = X + 1e-11 * np.arange(d)[None, :]
    a4, g4 = asp.ArrowSpaceBuilder.build(gp, Y)
    for i in (0, 1364, 1365, 12345, n - 1):
        assert np.array_equal(a4.get_item(i)[0], Y[i])
    Yd = torch.from_numpy(Y).cuda()
    a5, g5 = asp.ArrowSpaceBuilder.build_from_device(gp, Yd.data_ptr(), "float64", n, d, d)
    assert np.array_equal(a4.lambdas(), a5.lambdas())


@pytest.mark.parametrize("sharded", [False, True])
@pytest.mark.parametrize("n,d", [(130, 320), (700, 320), (1500, 256), (2900, 768), (5200, 320), (40000, 256)])
def test_every_row_is_scanned_whatever_the_grid(n, d, sharded):
    """The coarse tile scan hands its chunks out while it runs (scan_tile_kernel_dyn: groups of blocks, a cursor each).  Whatever
    the grid -- 1, 5, 11, 22, 40 or 312 blocks here: fewer blocks than sixteen groups once left chunks without a wave (a fuzz case
    on two ranks with shards of 115 and 142 rows found it) -- every row is read: each item, taken as the query, comes back as its
    own best hit (cosine 1), across the whole row range, on one space and on a row shard's one-exchange pass (one rank, no process
    group), under the scan that collects the scorer's candidates (tau = 0.62) and under the coarse chain's plain scan (tau = 0.2)."""
    import torch

    import pyarrowspace_amd as asp
    from pyarrowspace_amd.dist import ShardedIndex
    X = clustered(n, d, nclust=max(2, n // 400), seed=n)
    gp = {"eps": 1.25 * calibrate_eps(X, 8, "l2"), "k": 8, "topk": 5, "p": 2.0, "sigma": None}
    if sharded:
        index = ShardedIndex.build(gp, torch.from_numpy(X).cuda())
        search, operand = (lambda q, tau: index.search(q, tau)), (lambda: index.last_scan_operand())
    else:
        aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
        search, operand = (lambda q, tau: aspace.search(q, gl, tau)), (lambda: aspace.last_scan_operand)
    step = max(1, n // 260)
    ops = {0.62: set(), 0.2: set()}
    served = 0
    for tau in (0.62, 0.2):
        for i in list(range(0, n, step)) + [n - 1]:
            try:
                hits = search(np.ascontiguousarray(X[i]), tau)
            except asp.PanicException:      # (an item without a neighbour inside eps: the reference's assert)
                continue
            served += 1
            ops[tau].add(operand())
            if tau == 0.62:
                assert hits[0][0] == i, (n, d, tau, i, hits[:2])
            else:       # (the lambda term dominates at tau = 0.2: the item need not be among the hits)
                assert len(hits) == 5
    assert served >= 200, served
    if n >= 5000:
        assert "int8-high" in ops[0.62], ops     # (the scan under test did run; small shards may be served by the two-digit scan)
    if sharded:
        index.close()
