"""GPU parity of the feature mode (lambda_mode='feature', SPEC F1-F7): lambda as the reference's notes document it
-- Rayleigh quotient + edgewise dispersion on the F x F feature-space Laplacian (/root/reference/TAUMODE.md:8,12-27,
GRAPH_VARIABLES.md:17) -- against the fp64 CPU oracle, through the C ABI.  Bar: indices rank-exact, scores and
lambdas 1e-9 relative (north_star allows 1e-6)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import assert_hits_match, calibrate_feature_eps, clustered

pytestmark = pytest.mark.gpu

RTOL = 1e-9


def _gp(X, k, topk, metric, kernel, **extra):
    return dict({"eps": calibrate_feature_eps(X, k, metric), "k": k, "topk": topk, "p": 2.0, "sigma": None,
                 "metric": metric, "kernel": kernel, "lambda_mode": "feature"}, **extra)


def _check_feature_index(aspace, gl, ref):
    d = ref.X.shape[1]
    assert gl.lambda_mode == "feature" and gl.nnodes == d and gl.shape() == (d, d)
    assert aspace.nitems == ref.X.shape[0] and aspace.nfeatures == d
    indptr, indices, values = gl.to_csr()
    rows = np.repeat(np.arange(d), np.diff(indptr))
    off = indices != rows
    assert np.array_equal(indices[off], ref.indices)                       # feature-graph topology: exact
    np.testing.assert_allclose(values[off], ref.lap, rtol=RTOL, atol=1e-300)   # -w_ab
    np.testing.assert_allclose(values[~off], ref.deg, rtol=RTOL, atol=1e-300)  # L = D - W: the degree on the diagonal
    np.testing.assert_allclose(gl.degrees(), ref.deg, rtol=RTOL, atol=1e-300)
    assert abs(gl.tau0 - ref.tau0) <= RTOL * abs(ref.tau0)
    np.testing.assert_allclose(aspace.lambdas(), ref.lambdas, rtol=RTOL, atol=1e-300)
    # rows of a Laplacian sum to zero
    L = np.zeros((d, d))
    L[rows, indices] = values
    assert np.abs(L.sum(1)).max() <= 1e-12 * max(1.0, np.abs(values).max())


@pytest.mark.parametrize("metric,kernel", [("cosine", "rational"), ("cosine", "gaussian"), ("l2", "gaussian"), ("l2", "rational")])
@pytest.mark.parametrize("n,d,k,topk", [(300, 24, 5, 5), (1000, 384, 12, 10), (2500, 768, 25, 15), (37, 130, 4, 3), (5000, 200, 60, 20)])
def test_feature_build_and_search_match_oracle(oracle_lib, n, d, k, topk, metric, kernel):
    """Shapes: D below / across / at multiples of the 128-wide Gram tile, N below one 32-row stage, k above the
    item graph's cap of 56 (the feature graph ranks whole columns)."""
    import pyarrowspace_amd as asp
    X = clustered(n, d, nclust=max(4, n // 64), seed=n + d)
    gp = _gp(X, k, topk, metric, kernel)
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    ref = oracle_lib.OracleIndex(X, gp)
    _check_feature_index(aspace, gl, ref)
    rng = np.random.default_rng(7)
    for qi in range(4):
        q = X[rng.integers(0, n)] + 0.05 * rng.standard_normal(d) / np.sqrt(d)
        lq_ref = ref.query_lambda(q)
        assert abs(aspace.query_lambda(q, gl) - lq_ref) <= RTOL * abs(lq_ref)
        for tau in (1.0, 0.62, 0.0):
            want, _ = ref.search(q, tau)
            got = aspace.search(q, gl, tau)
            assert_hits_match(got, want, ref.scores(q, tau, lq_ref), rtol=RTOL)


def test_feature_lambda_is_scale_invariant_in_the_query(oracle_lib):
    """E and G are ratios of quadratic forms: q and c*q have the same lambda, so a scaled copy of an item always
    ranks that item first -- the reason tests/test_0.py:39-61 cannot hold under any documented lambda."""
    import pyarrowspace_amd as asp
    X = clustered(400, 48, nclust=6, seed=2)
    gp = _gp(X, 6, 3, "cosine", "rational")
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    lam = aspace.lambdas()
    for c in (1.05, 0.5, 4.0):   # powers of two are exact, 1.05 to rounding
        lq = aspace.query_lambda(np.ascontiguousarray(X[7] * c), gl)
        assert abs(lq - lam[7]) <= 1e-12 * lam[7]
        for tau in (0.9, 0.55, 0.0):
            assert aspace.search(np.ascontiguousarray(X[7] * c), gl, tau)[0][0] == 7


def test_feature_items_kept_in_fp64(oracle_lib):
    """Items that do not round-trip through fp32: the Gram and the energies read the fp64 copy."""
    import pyarrowspace_amd as asp
    X = clustered(700, 72, nclust=8, seed=9) * (1.0 + 1e-9)
    assert not np.array_equal(X.astype(np.float32).astype(np.float64), X)
    gp = _gp(X, 7, 6, "cosine", "rational")
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    ref = oracle_lib.OracleIndex(X, gp)
    _check_feature_index(aspace, gl, ref)


def test_feature_degenerate_inputs(oracle_lib):
    """A zero column (cosine 0 to everything), duplicate columns (distance exactly 0, ordered by index), a zero
    item (lambda 0) and a query that is constant over every connected pair of features (lambda_q == 0 -> panic)."""
    import pyarrowspace_amd as asp
    X = clustered(300, 20, nclust=5, seed=4)
    X[:, 3] = 0.0
    X[:, 11] = X[:, 5]
    X[:, 17] = X[:, 5]
    X[42] = 0.0
    gp = _gp(X, 4, 3, "cosine", "rational")
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    ref = oracle_lib.OracleIndex(X, gp)
    _check_feature_index(aspace, gl, ref)
    assert aspace.lambdas()[42] == 0.0
    with pytest.raises(asp.PanicException):
        aspace.search(np.ones(20), gl, 0.5)
    with pytest.raises(oracle_lib.ZeroLambda):
        ref.search(np.ones(20), 0.5)


def test_feature_search_batch_matches_single(oracle_lib):
    import pyarrowspace_amd as asp
    n, d = 3000, 200
    X = clustered(n, d, nclust=10, seed=13)
    gp = _gp(X, 9, 7, "cosine", "rational")
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    ref = oracle_lib.OracleIndex(X, gp)
    rng = np.random.default_rng(17)
    Q = np.stack([X[rng.integers(0, n)] + 0.03 * rng.standard_normal(d) / np.sqrt(d) for _ in range(45)])
    got = aspace.search_batch(Q, gl, 0.62)
    for b in range(45):
        want, lq = ref.search(Q[b], 0.62)
        assert_hits_match(got[b], want, ref.scores(Q[b], 0.62, lq), rtol=RTOL)
        assert got[b] == aspace.search(np.ascontiguousarray(Q[b]), gl, 0.62)


def test_feature_index_save_load_roundtrip(oracle_lib, tmp_path):
    import pyarrowspace_amd as asp
    X = clustered(800, 96, nclust=8, seed=21)
    gp = _gp(X, 8, 5, "cosine", "rational")
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    path = str(tmp_path / "feat.asidx")
    aspace.save(gl, path)
    a2, g2 = asp.ArrowSpaceBuilder.load(path)
    assert g2.lambda_mode == "feature" and g2.nnodes == 96 and g2.graph_params == gl.graph_params
    np.testing.assert_array_equal(a2.lambdas(), aspace.lambdas())
    for t1, t2 in zip(gl.to_csr(), g2.to_csr()):
        np.testing.assert_array_equal(t1, t2)
    q = np.ascontiguousarray(X[3] * 1.01 + 0.001)
    assert a2.search(q, g2, 0.62) == aspace.search(q, gl, 0.62)
    # a graph handle of another space is refused
    b, gb = asp.ArrowSpaceBuilder.build(_gp(X[:100], 8, 5, "cosine", "rational"), X[:100])
    with pytest.raises(ValueError):
        aspace.search(q, gb, 0.62)


def test_feature_staged_steps_compose_to_the_fused_build(oracle_lib):
    """as_feat_gram over two row ranges + as_feat_graph + as_feat_energy per range + as_feat_lambdas == as_build
    (what a row-sharded multi-GPU host does)."""
    import torch

    import pyarrowspace_amd as asp
    from pyarrowspace_amd import _lib
    L = _lib.load()
    n, d = 1200, 160
    X = clustered(n, d, nclust=8, seed=31)
    gpd = _gp(X, 10, 5, "cosine", "rational")
    fused, gl = asp.ArrowSpaceBuilder.build(gpd, X)
    gp, op = asp._parse_graph_params(gpd)
    Xd = torch.from_numpy(X).cuda()
    sp = C.c_void_p()
    assert L.as_space_create_dev(C.c_void_p(Xd.data_ptr()), _lib.DTYPE_F64, n, d, d, C.byref(op), C.byref(sp)) == 0
    g1 = torch.empty((d, d), dtype=torch.float64, device="cuda")
    g2 = torch.empty_like(g1)
    assert L.as_feat_gram(sp, 0, 500, C.c_void_p(g1.data_ptr())) == 0
    assert L.as_feat_gram(sp, 500, n, C.c_void_p(g2.data_ptr())) == 0
    gram = g1 + g2
    np.testing.assert_allclose(gram.cpu().numpy(), X.T @ X, rtol=1e-12, atol=1e-12)
    gr = C.c_void_p()
    assert L.as_feat_graph(sp, C.byref(gp), C.c_void_p(gram.data_ptr()), C.byref(gr)) == 0
    E = torch.zeros(n, dtype=torch.float64, device="cuda")
    G = torch.zeros(n, dtype=torch.float64, device="cuda")
    assert L.as_feat_energy(sp, gr, 0, 700, C.c_void_p(E.data_ptr()), C.c_void_p(G.data_ptr())) == 0
    assert L.as_feat_energy(sp, gr, 700, n, C.c_void_p(E.data_ptr()), C.c_void_p(G.data_ptr())) == 0
    assert L.as_feat_lambdas(sp, gr, C.c_void_p(E.data_ptr()), C.c_void_p(G.data_ptr())) == 0
    lam = np.empty(n)
    assert L.as_lambdas(sp, lam.ctypes.data_as(C.c_void_p)) == 0
    np.testing.assert_allclose(lam, fused.lambdas(), rtol=1e-12)
    L.as_free_graph(gr)
    L.as_free_space(sp)
