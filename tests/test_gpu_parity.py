"""GPU parity: the HIP path (through the C ABI, via the arrowspace-compatible host
module) against the fp64 CPU oracle on the same seeded inputs.  Bar (BASELINE.json
north_star): returned indices rank-exact, fp64 scores within 1e-6 relative (we assert
1e-9), lambdas within 1e-9 relative."""
import numpy as np
import pytest

from conftest import assert_hits_match, calibrate_eps, clustered

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def _build_both(X, gp, oracle_lib):
    import pyarrowspace_amd as asp
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    ref = oracle_lib.OracleIndex(X, gp)
    return aspace, gl, ref


def _check_index(aspace, gl, ref):
    lam = aspace.lambdas()
    np.testing.assert_allclose(lam, ref.lambdas, rtol=RTOL, atol=1e-300)
    np.testing.assert_allclose(gl.degrees(), ref.deg, rtol=RTOL, atol=1e-300)
    assert abs(gl.tau0 - ref.tau0) <= RTOL * abs(ref.tau0)
    indptr, indices, values = gl.to_csr()
    n = ref.X.shape[0]
    # oracle CSR without the diagonal -> strip ours
    rows = np.repeat(np.arange(n), np.diff(indptr))
    off = indices != rows
    assert np.array_equal(indices[off], ref.indices)
    np.testing.assert_allclose(values[off], ref.lap, rtol=RTOL, atol=1e-300)
    np.testing.assert_array_equal(values[~off], (ref.deg > 0).astype(np.float64))


@pytest.mark.parametrize("metric,kernel", [("l2", "gaussian"), ("cosine", "rational"), ("l2", "rational"), ("cosine", "gaussian")])
@pytest.mark.parametrize("n,d,k,topk", [(300, 24, 6, 5), (1000, 384, 12, 10), (2500, 768, 25, 15)])
def test_build_and_search_match_oracle(oracle_lib, n, d, k, topk, metric, kernel):
    X = clustered(n, d, nclust=max(4, n // 64), seed=n + d)
    eps = calibrate_eps(X, k, metric)
    gp = {"eps": eps, "k": k, "topk": topk, "p": 2.0, "sigma": None, "metric": metric, "kernel": kernel}
    aspace, gl, ref = _build_both(X, gp, oracle_lib)
    _check_index(aspace, gl, ref)
    rng = np.random.default_rng(7)
    for qi in range(6):
        q = X[rng.integers(0, n)] + 0.05 * rng.standard_normal(d) / np.sqrt(d)
        for tau in (1.0, 0.8, 0.62, 0.42, 0.0):
            try:
                want, lq_ref = ref.search(q, tau)
            except oracle_lib.ZeroLambda:
                import pyarrowspace_amd as asp
                with pytest.raises(asp.PanicException):
                    aspace.search(q, gl, tau)
                continue
            got = aspace.search(q, gl, tau)
            assert_hits_match(got, want, ref.scores(q, tau, lq_ref), rtol=RTOL)
            assert abs(aspace.query_lambda(q, gl) - lq_ref) <= RTOL * abs(lq_ref)


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_search_fallback_paths_match_oracle(oracle_lib, mode):
    """bit0 = fp64 end to end, bit1 = wavefront-shuffle list selection (overflow fallback)."""
    import pyarrowspace_amd as asp
    n, d, k, topk = 1500, 96, 10, 8
    X = clustered(n, d, nclust=12, seed=11)
    eps = calibrate_eps(X, k)
    gp = {"eps": eps, "k": k, "topk": topk, "p": 2.0, "sigma": None, "_search_mode": mode}
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    ref = oracle_lib.OracleIndex(X, gp)
    rng = np.random.default_rng(3)
    for _ in range(5):
        q = X[rng.integers(0, n)] + 0.02 * rng.standard_normal(d) / np.sqrt(d)
        for tau in (1.0, 0.62, 0.0):
            want, lq = ref.search(q, tau)
            got = aspace.search(q, gl, tau)
            assert_hits_match(got, want, ref.scores(q, tau, lq), rtol=RTOL)   # ties to rounding (tau = 0) may swap


def test_build_exact_fallback_matches_oracle(oracle_lib):
    """force_exact routes every row through the fp64 brute-force fallback of the build."""
    n, d, k = 400, 40, 7
    X = clustered(n, d, nclust=6, seed=5)
    gp = {"eps": calibrate_eps(X, k), "k": k, "topk": 5, "p": 2.0, "sigma": None, "force_exact": True}
    aspace, gl, ref = _build_both(X, gp, oracle_lib)
    _check_index(aspace, gl, ref)


def test_non_fp32_representable_items_keep_fp64(oracle_lib):
    """Items that do not round-trip through fp32 keep an fp64 copy for the exact re-evaluation."""
    n, d, k = 600, 48, 6
    X = clustered(n, d, nclust=8, seed=9) * (1.0 + 1e-9)
    assert not np.array_equal(X.astype(np.float32).astype(np.float64), X)
    gp = {"eps": calibrate_eps(X, k), "k": k, "topk": 6, "p": 2.0, "sigma": None}
    aspace, gl, ref = _build_both(X, gp, oracle_lib)
    _check_index(aspace, gl, ref)
    v, lam = aspace.get_item(17)
    assert np.array_equal(v, X[17]) and abs(lam - ref.lambdas[17]) <= RTOL * abs(ref.lambdas[17])
    q = X[5] * 1.01
    want, _ = ref.search(q, 0.62)
    got = aspace.search(q, gl, 0.62)
    assert [i for i, _ in got] == [i for i, _ in want]


@pytest.mark.parametrize("metric,d", [("l2", 200), ("cosine", 200), ("l2", 24), ("cosine", 768), ("l2", 1000), ("l2", 1024), ("cosine", 1536),
                                      ("l2", 2048), ("cosine", 3000), ("l2", 4096)])
def test_search_batch_matches_single_and_oracle(oracle_lib, metric, d):
    """as_search_batch: 32 query slots per pass over the items (MFMA pass, K split over 4 waves: 7 slabs -> 2,2,2,1;
    1 slab -> 1,0,0,0; 24 slabs -> 6 each; rows wider than 768 floats on the 8-slabs-per-wave instantiation: 1 024 columns in one
    launch, beyond that K-chunk passes -- 1536 -> 2 x 768, 2048 -> 2 x 1024, 3008 -> 1024 / 1024 / 960, 4096 -> 4 x 1024 -- whose
    partial dots gemm_combine_kernel adds), chunks of 32 and 13 queries."""
    import pyarrowspace_amd as asp
    n, k, topk = 3000, 9, 7
    X = clustered(n, d, nclust=10, seed=13)
    gp = {"eps": calibrate_eps(X, k, metric), "k": k, "topk": topk, "p": 2.0, "sigma": None, "metric": metric}
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    ref = oracle_lib.OracleIndex(X, gp)
    rng = np.random.default_rng(17)
    Q = np.stack([X[rng.integers(0, n)] + 0.03 * rng.standard_normal(d) / np.sqrt(d) for _ in range(45)])
    for tau in (0.62, 1.0):
        got = aspace.search_batch(Q, gl, tau)
        assert len(got) == 45
        assert aspace.last_batch_int8          # clustered rows: the int8 images serve the pass (three i8 products per column)
        for b in range(45):
            want, lq = ref.search(Q[b], tau)
            assert_hits_match(got[b], want, ref.scores(Q[b], tau, lq), rtol=RTOL)
            assert got[b] == aspace.search(np.ascontiguousarray(Q[b]), gl, tau)
            assert aspace.last_scan_int8          # rows of up to 4 096 floats: the single-query scan reads the image too
    far = Q.copy()
    far[4] = 0.0
    far[4, 0] = 40.0                      # one query without neighbours poisons the batch like the reference's assert
    with pytest.raises(asp.PanicException):
        aspace.search_batch(far, gl, 0.62)


@pytest.mark.parametrize("metric,d", [("l2", 768), ("cosine", 320)])
def test_search_batch_pairs_share_one_scan(oracle_lib, metric, d):
    """as_search_batch, more than 32 queries: the passes are launched in PAIRS and a pair on the int8 images (rows of up to 768
    columns) is served by ONE scan for its 64 queries (scan_gemm_dual_kernel); more than 64 queries: two pairs of workspaces
    alternate.  200 queries (three pairs and a single pass of 8), 64 (one pair), 45 (32 + 13): every slot is what the single
    search returns and what the oracle returns; with ARROWSPACE_NO_BATCH_DUAL=1 (each workspace scans for itself) the same."""
    import os

    import pyarrowspace_amd as asp
    n, k, topk = 5000, 9, 7
    X = clustered(n, d, nclust=12, seed=29)
    gp = {"eps": calibrate_eps(X, k, metric), "k": k, "topk": topk, "p": 2.0, "sigma": None, "metric": metric}
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    ref = oracle_lib.OracleIndex(X, gp)
    rng = np.random.default_rng(19)
    Q = np.stack([X[rng.integers(0, n)] + 0.03 * rng.standard_normal(d) / np.sqrt(d) for _ in range(200)])
    single = [aspace.search(np.ascontiguousarray(q), gl, 0.62) for q in Q]
    for b in range(0, 200, 9):
        want, lq = ref.search(Q[b], 0.62)
        assert_hits_match(single[b], want, ref.scores(Q[b], 0.62, lq), rtol=RTOL)
    before = aspace.batch_dual_scans
    for nq, pairs in ((200, 3), (64, 1), (45, 1), (33, 1), (32, 0)):
        got = aspace.search_batch(Q[:nq], gl, 0.62)
        assert aspace.last_batch_int8
        assert got == single[:nq], nq
        assert aspace.batch_dual_scans - before == pairs, (nq, aspace.batch_dual_scans - before)
        before = aspace.batch_dual_scans
    os.environ["ARROWSPACE_NO_BATCH_DUAL"] = "1"     # (read once per process: a child decides)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import pyarrowspace_amd as asp\n"
            "d = np.load(sys.argv[1]); gp = dict(eps=float(d['eps']), k=%d, topk=%d, p=2.0, sigma=None, metric=%r)\n"
            "a, gl = asp.ArrowSpaceBuilder.build(gp, d['X']); got = a.search_batch(d['Q'], gl, 0.62)\n"
            "assert a.batch_dual_scans == 0\n"
            "np.save(sys.argv[2], np.array([[i for i, _ in h] for h in got]))\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), k, topk, metric)
    import subprocess
    import sys
    import tempfile
    try:
        with tempfile.TemporaryDirectory() as tmp:
            np.savez(os.path.join(tmp, "in.npz"), X=X, Q=Q[:100], eps=gp["eps"])
            out = subprocess.run([sys.executable, "-c", code, os.path.join(tmp, "in.npz"), os.path.join(tmp, "out.npy")], capture_output=True, text=True, timeout=600)
            assert out.returncode == 0, out.stderr[-2000:]
            idx = np.load(os.path.join(tmp, "out.npy"))
    finally:
        os.environ.pop("ARROWSPACE_NO_BATCH_DUAL", None)
    assert [list(r) for r in idx] == [[i for i, _ in h] for h in single[:100]]


def test_int8_image_scan_returns_what_the_fp32_scan_returns(oracle_lib):
    """The single-query scan reads the int8 two-digit image of the items (half the bytes) when the items' and the query's
    quantisation error allow it; it only prefilters -- k-NN candidates and scorer candidates are re-evaluated exactly and proven
    against the wider error term -- so hits, scores and lambda_q are those of the fp32 scan (ARROWSPACE_SCAN_FP32=1) and of the
    oracle: clustered items, three taus, queries near items, an exact item, a scaled item, a query with one dominant component."""
    import os

    import pyarrowspace_amd as asp
    n, d, k, topk = 8000, 200, 12, 9
    X = clustered(n, d, nclust=10, seed=31)
    gp = {"eps": calibrate_eps(X, k, "l2"), "k": k, "topk": topk, "p": 2.0, "sigma": None}
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    assert aspace.knn_pipe == "int8"
    ref = oracle_lib.OracleIndex(X, gp)
    rng = np.random.default_rng(3)
    Q = [np.ascontiguousarray(X[rng.integers(0, n)] + 0.03 * rng.standard_normal(d) / np.sqrt(d)) for _ in range(24)]
    Q.append(np.ascontiguousarray(X[77]))
    Q.append(np.ascontiguousarray(3.5 * X[1234]))
    spike = np.ascontiguousarray(X[5] * 1e-3)
    spike[0] = 1.0                      # one dominant component: s_q / |q| ~ 1 -- the query's own residue term is at its largest
    Q.append(spike)
    used = []
    for tau in (0.62, 1.0, 0.0):
        for q in Q:
            os.environ.pop("ARROWSPACE_SCAN_FP32", None)
            try:
                got = aspace.search(q, gl, tau)
                used.append(aspace.last_scan_int8)
            except asp.PanicException:
                got = None
            os.environ["ARROWSPACE_SCAN_FP32"] = "1"
            try:
                try:
                    want32 = aspace.search(q, gl, tau)
                    assert not aspace.last_scan_int8
                except asp.PanicException:
                    want32 = None
            finally:
                os.environ.pop("ARROWSPACE_SCAN_FP32", None)
            assert got == want32
            if got is not None:
                want, lq = ref.search(q, tau)
                assert_hits_match(got, want, ref.scores(q, tau, lq), rtol=RTOL)
    assert all(used) and len(used) >= 70
    with pytest.raises(asp.PanicException):
        aspace.search(np.zeros(d), gl, 0.62)
    assert not aspace.last_scan_int8    # a query without a scale (all zeros) is the fp32 scan's
    c = aspace.search_counters()
    assert c["searches_with_rerun"] == 0


def test_batched_int8_pass_holds_measured_residues_against_the_assumed_ones(oracle_lib):
    """The batched int8 pass is priced BEFORE its launch with residue norms of the queries the host assumes (1.25 x what earlier
    passes measured); the device measures the real ones and the host compares when it collects.  A batch that breaks the
    assumption -- queries with one dominant component after passes of smooth ones: s_q / |q| near 1 instead of 0.2 -- must still
    return the oracle's hits (its slots are rerun singly), and so must the smooth batches before and after it."""
    import pyarrowspace_amd as asp
    n, d, k, topk = 4000, 256, 10, 8
    X = clustered(n, d, nclust=10, seed=61)
    gp = {"eps": calibrate_eps(X, k, "cosine"), "k": k, "topk": topk, "p": 2.0, "sigma": None, "metric": "cosine"}
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    ref = oracle_lib.OracleIndex(X, gp)
    rng = np.random.default_rng(5)
    smooth = np.stack([X[rng.integers(0, n)] + 0.03 * rng.standard_normal(d) / np.sqrt(d) for _ in range(40)])
    spiky = smooth.copy()
    for b in range(0, 40, 3):
        spiky[b] = 0.02 * spiky[b]
        spiky[b, b % d] = 1.0

    def check(Q, tau):
        got = aspace.search_batch(Q, gl, tau)
        for b in range(len(Q)):
            try:
                want, lq = ref.search(Q[b], tau)
            except oracle_lib.ZeroLambda:
                continue
            assert_hits_match(got[b], want, ref.scores(Q[b], tau, lq), rtol=RTOL)

    for tau in (0.62, 1.0):
        check(smooth, tau)
        assert aspace.last_batch_int8
        try:
            check(spiky, tau)
        except asp.PanicException:            # (a spiky query without a neighbour poisons the batch like the reference's assert)
            for b in range(len(spiky)):
                try:
                    want, lq = ref.search(spiky[b], tau)
                except oracle_lib.ZeroLambda:
                    continue
                assert_hits_match(aspace.search(np.ascontiguousarray(spiky[b]), gl, tau), want, ref.scores(spiky[b], tau, lq), rtol=RTOL)
        check(smooth, tau)


@pytest.mark.parametrize("metric", ["l2", "cosine"])
def test_coarse_scan_returns_what_the_two_digit_scan_returns(oracle_lib, metric):
    """tau >= 0.4 on a single space: the scan reads the HIGH digits of the int8 image alone (1 B per element) -- a prefilter off by
    up to V |x||q| (1e-2) -- and every k-NN candidate and every scorer candidate it keeps is re-evaluated exactly by the tail's
    blocks; same hits, bit for bit, as the two-digit scan (ARROWSPACE_SCAN_COARSE=0) and as the oracle.  tau below 0.4 takes the
    coarse CHAIN since round 5 (no cosine window there: the scorer's candidates by threshold over the kept coarse dots once
    lambda_q is known, every one of them evaluated exactly) -- same hits again; a query whose candidates do not fit is redone
    inside the same call."""
    import os

    import pyarrowspace_amd as asp
    n, d, k, topk = 30000, 256, 12, 9
    X = clustered(n, d, nclust=150, seed=71)
    gp = {"eps": calibrate_eps(X, k, metric), "k": k, "topk": topk, "p": 2.0, "sigma": None, "metric": metric}
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    ref = oracle_lib.OracleIndex(X, gp)
    rng = np.random.default_rng(11)
    Q = [np.ascontiguousarray(X[rng.integers(0, n)] + 0.03 * rng.standard_normal(d) / np.sqrt(d)) for _ in range(20)]
    Q.append(np.ascontiguousarray(X[123]))
    ops = []
    try:
        for tau in (0.62, 1.0, 0.4, 0.2):
            for q in Q:
                os.environ["ARROWSPACE_SCAN_COARSE"] = "2"       # (every search probes the coarse scan, whatever the last one did)
                try:
                    got = aspace.search(q, gl, tau)
                except asp.PanicException:                       # (no item inside eps: the reference's assert, on either scan)
                    got = None
                ops.append((tau, aspace.last_scan_operand))
                os.environ["ARROWSPACE_SCAN_COARSE"] = "0"
                try:
                    fine = aspace.search(q, gl, tau)
                except asp.PanicException:
                    fine = None
                assert aspace.last_scan_operand == "int8"
                assert got == fine
                try:
                    want, lq = ref.search(q, tau)
                except oracle_lib.ZeroLambda:
                    assert got is None
                    continue
                assert_hits_match(got, want, ref.scores(q, tau, lq), rtol=RTOL)
    finally:
        os.environ.pop("ARROWSPACE_SCAN_COARSE", None)
    assert sum(op == "int8-high" for tau, op in ops if tau < 0.4) >= 3, ops      # (the coarse chain)
    coarse = [op == "int8-high" for tau, op in ops if tau >= 0.4]
    # (a query whose coarse candidates do not fit -- on an index this small the bound is learnt late, and how late is a matter
    # of timing -- is redone on the two-digit image inside the call and reports "int8")
    assert sum(coarse) >= 3, ops
    assert aspace.search_counters()["searches_with_rerun"] == 0
