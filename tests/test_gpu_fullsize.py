"""GPU, BASELINE.json's headline size (N = 1M x D = 768 fp32, k = 25, topk = 15), in the north_star mode (L2 distance,
Gaussian weights) and in the mode the reference's own parameter sets are written for (rectified-cosine distance,
rational weights: GRAPH_VARIABLES.md:7-10): the oracle cannot run an all-pairs build at this size, so parity is
checked through size-independent properties -- exact k-NN of sampled rows against an independent fp64 brute force
(torch), Laplacian identities (symmetry, L D^1/2 1 = 0), score self-consistency (TAUMODE.md:33), sortedness,
determinism of a rebuild, batch == single.  tests/test_gpu_configs.py does the same for BASELINE.json's other configs."""
import os
import sys

import numpy as np
import pytest

from conftest import brute_keys, gpu_clustered

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
N, D, K, TOPK, TAU = 1_000_000, 768, 25, 15, 0.62


def check_laplacian(csr, deg, n, k):
    """Shape, ascending columns, unit / zero diagonal, negative off-diagonal, symmetry, L D^1/2 1 = 0."""
    indptr, indices, values = csr
    assert indptr.shape == (n + 1,) and indptr[-1] == len(indices)
    rowlen = np.diff(indptr)
    assert rowlen.min() >= 1 and (rowlen - 1).sum() <= 2 * n * k      # diagonal + at most k out- and k*? in-edges
    rows = np.repeat(np.arange(n), rowlen)
    diag = indices == rows
    assert diag.sum() == n
    np.testing.assert_array_equal(values[diag], (deg > 0).astype(np.float64))
    assert (values[~diag] < 0).all() and (np.diff(indices)[np.diff(rows) == 0] > 0).all()
    # symmetry of pattern and values: sum_ij L_ij u_i v_j == sum_ij L_ij v_i u_j for random u, v
    rng = np.random.default_rng(0)
    u, v = rng.standard_normal(n), rng.standard_normal(n)
    a = np.sum(values * u[rows] * v[indices])
    b = np.sum(values * v[rows] * u[indices])
    assert abs(a - b) <= 1e-9 * (abs(a) + abs(b) + 1)
    r = np.bincount(rows, weights=values * np.sqrt(deg[indices]), minlength=n)
    assert np.max(np.abs(r[deg > 0])) < 1e-9


def check_sampled_knn(X, csr, metric, eps, k, nsample=64, seed=1, rel=1e-9):
    """Sampled rows: every item the fp64 brute force puts safely inside (eps, k-th distance) is a neighbour;
    nothing safely outside eps is."""
    import torch
    indptr, indices, _ = csr
    n = X.shape[0]
    epskey = eps * eps if metric == "l2" else eps
    sample = np.random.default_rng(seed).choice(n, nsample, replace=False)
    keys = brute_keys(X, sample, metric)
    vals, idx = torch.topk(keys, k + 8, dim=1, largest=False)
    vals, idx = vals.cpu().numpy(), idx.cpu().numpy()
    scale = float((X[:4096].double() ** 2).sum(1).max().item()) if metric == "l2" else 1.0
    margin = rel * scale
    for t, i in enumerate(sample):
        cols = indices[indptr[i]:indptr[i + 1]]
        cols = set(cols[cols != i].tolist())
        inside = [int(j) for v, j in zip(vals[t][:k], idx[t][:k]) if v <= epskey - margin and (vals[t][k] - v) > margin]
        assert set(inside) <= cols, (i, set(inside) - cols)
        if np.isfinite(epskey):
            for v, j in zip(vals[t], idx[t]):   # a column outside eps can never be an edge, in either direction
                assert not (v > epskey + margin and int(j) in cols), (i, int(j), v)


def check_search(X, aspace, gl, lam, tau, topk, rows, seed=2):
    """TAUMODE.md:33 recomputed from the accessors; sortedness; no left-out item beats the last hit; batch == single."""
    import torch
    n, d = X.shape
    rng = np.random.default_rng(seed)
    for i in rows:
        x = X[int(i)].double().cpu().numpy()
        q = np.ascontiguousarray(x * 1.01 + 0.002 * np.linalg.norm(x) * rng.standard_normal(d) / np.sqrt(d))
        hits = aspace.search(q, gl, tau)
        lq = aspace.query_lambda(q, gl)
        sc = [s for _, s in hits]
        assert len(hits) == topk and sc == sorted(sc, reverse=True) and len(set(j for j, _ in hits)) == topk
        for j, s in hits:
            v, lj = aspace.get_item(j)
            cos = float(v @ q) / np.sqrt(float(v @ v) * float(q @ q))
            assert abs(s - (tau * cos + (1 - tau) / (1 + abs(lq - lj)))) < 1e-12 and lj == lam[j]
        qd = torch.from_numpy(q).cuda()
        Xd = X.double()
        cosall = (Xd @ qd) / torch.sqrt((Xd * Xd).sum(1) * (qd @ qd))
        del Xd
        sall = tau * cosall + (1 - tau) / (1 + torch.abs(lq - torch.from_numpy(lam).cuda()))
        top = torch.topk(sall, topk).values.cpu().numpy()
        np.testing.assert_allclose(sc, top, rtol=1e-9)
        assert aspace.search_batch(np.stack([q, x]), gl, tau)[0] == hits


@pytest.fixture(scope="module", params=[("l2", "gaussian"), ("cosine", "rational")], ids=["l2-gaussian", "cosine-rational"])
def big(request):
    import torch

    import bench
    import pyarrowspace_amd as asp
    metric, kernel = request.param
    X = gpu_clustered(N, D, 42)
    eps = bench.calibrate_eps(X, K, metric)
    gp = {"eps": eps, "k": K, "topk": TOPK, "p": 2.0, "sigma": None, "metric": metric, "kernel": kernel}
    aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", N, D, D)
    yield dict(X=X, gp=gp, metric=metric, aspace=aspace, gl=gl, lam=aspace.lambdas(), deg=gl.degrees(), csr=gl.to_csr(), torch=torch)
    del aspace, gl, X
    torch.cuda.empty_cache()


def test_graph_shape_and_laplacian_identities(big):
    check_laplacian(big["csr"], big["deg"], N, K)
    stats = big["gl"].build_stats()
    assert stats["fallback_rows"] <= N * 0.01
    assert stats["mfma_flops"] / stats["knn_mfma_s"] > 0.5 * 157.3e12       # the X.X^T block keeps north_star's >= 50 % of fp32 MFMA peak
    assert 0 < big["gl"].tau0 <= 1 and np.isfinite(big["lam"]).all() and (big["lam"] >= 0).all()


def test_sampled_rows_have_exact_knn(big):
    check_sampled_knn(big["X"], big["csr"], big["metric"], big["gp"]["eps"], K)


def test_search_properties(big):
    X, aspace, gl = big["X"], big["aspace"], big["gl"]
    rows = np.random.default_rng(2).choice(N, 5, replace=False)
    for i in rows[:2]:
        x = X[int(i)].double().cpu().numpy()
        hits = aspace.search(x, gl, 1.0)
        assert len(hits) == TOPK and hits[0][0] == int(i) and abs(hits[0][1] - 1.0) < 1e-12
    check_search(X, aspace, gl, big["lam"], TAU, TOPK, rows)


def test_rebuild_is_bitwise_deterministic(big):
    import pyarrowspace_amd as asp
    X = big["X"]
    a2, g2 = asp.ArrowSpaceBuilder.build_from_device(big["gp"], X.data_ptr(), "float32", N, D, D)
    assert np.array_equal(a2.lambdas(), big["lam"]) and g2.tau0 == big["gl"].tau0


def test_feature_mode_at_headline_size():
    """lambda_mode='feature' at 1M x 768 (SPEC F1-F7): sampled Gram-derived edges, L = D - W identities, lambdas of
    sampled items recomputed in fp64 from the CSR (x^T L x / x^T x and the dispersion), search properties."""
    import torch

    import bench
    import pyarrowspace_amd as asp
    X = gpu_clustered(N, D, 42)
    eps = bench.calibrate_feature_eps(X, K, "cosine")
    gp = {"eps": eps, "k": K, "topk": TOPK, "p": 2.0, "sigma": None, "metric": "cosine", "kernel": "rational", "lambda_mode": "feature"}
    aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", N, D, D)
    assert gl.nnodes == D and aspace.nitems == N and gl.lambda_mode == "feature"
    indptr, indices, values = gl.to_csr()
    deg = gl.degrees()
    rows = np.repeat(np.arange(D), np.diff(indptr))
    L = np.zeros((D, D))
    L[rows, indices] = values
    assert np.allclose(L, L.T, rtol=0, atol=0) and np.abs(L.sum(1)).max() < 1e-9 and np.array_equal(np.diag(L), deg)
    W = -(L - np.diag(deg))
    assert (W >= 0).all() and ((W > 0).sum(1) >= 1).all()
    # edges against an independent fp64 Gram (torch): every edge inside eps, weights = 1 / (1 + (d / sigma)^2)
    G = torch.zeros((D, D), dtype=torch.float64, device="cuda")
    for s in range(0, N, 1 << 16):
        B = X[s:s + (1 << 16)].double()
        G += B.T @ B
    G = G.cpu().numpy()
    m = np.diag(G)
    dist = 1.0 - np.clip(G / np.sqrt(np.outer(m, m)), 0, 1)
    a, b = np.nonzero(W)
    assert (dist[a, b] <= eps + 1e-12).all()
    np.testing.assert_allclose(W[a, b], 1.0 / (1.0 + (dist[a, b] / (eps * 0.5)) ** 2), rtol=1e-9)
    kth = np.sort(dist + np.eye(D) * 9, axis=1)[:, K - 1]          # k nearest of every column are edges
    for c in range(0, D, 37):
        near = np.nonzero((dist[c] < min(kth[c], eps) - 1e-9) & (np.arange(D) != c))[0]
        assert set(near.tolist()) <= set(np.nonzero(W[c])[0].tolist())
    # lambdas of sampled items from the definition
    lam, tau0 = aspace.lambdas(), gl.tau0
    sample = np.random.default_rng(3).choice(N, 256, replace=False)
    Xs = X[torch.from_numpy(sample).cuda()].double().cpu().numpy()
    E = np.einsum("ic,cd,id->i", Xs, L, Xs) / np.einsum("ic,ic->i", Xs, Xs)
    ua, ub = np.nonzero(np.triu(W, 1))
    e = W[ua, ub][None, :] * (Xs[:, ua] - Xs[:, ub]) ** 2
    Gd = ((e / e.sum(1, keepdims=True)) ** 2).sum(1)
    np.testing.assert_allclose(lam[sample], tau0 * E / (E + tau0) + (1 - tau0) * Gd, rtol=1e-9)
    assert 0 < tau0 <= 1 and np.isfinite(lam).all()
    check_search(X, aspace, gl, lam, TAU, TOPK, sample[:3])
    stats = gl.build_stats()
    assert stats["total_s"] < 2.0          # N-linear: no all-pairs work in this mode


def test_coarse_scan_against_the_oracle_at_headline_size(big):
    """The headline's own path -- the coarse scan over the tiles of the int8 image's high digits, the merged exact tail -- held
    against the ORACLE (C restatement, fp64 over all 1M items) on 216 queries: tau in {0.4, 0.62, 1.0}, half of them perturbed
    items (the bench's timed queries), half fresh draws around the index's centres (SURVEY 8(d)'s query recipe; the ones with no
    item inside eps must raise the zero-lambda panic on both sides).  Indices rank-exact, scores and lambda_q to 1e-9."""
    import torch

    import pyarrowspace_amd as asp
    from conftest import assert_hits_match
    from oracle import oracle_c
    X, aspace, gl, gp = big["X"], big["aspace"], big["gl"], big["gp"]
    ref = oracle_c.OracleSearchOnly(X.double().cpu().numpy(), gp, big["deg"], big["lam"], gl.tau0)
    g = torch.Generator(device="cuda")
    g.manual_seed(42)
    C = torch.randn((1024, D), generator=g, device="cuda", dtype=torch.float32).double().cpu().numpy()   # gpu_clustered's centres
    rng = np.random.default_rng(77)
    per = 36
    operands, zero = set(), 0
    for tau in (0.4, 0.62, 1.0):
        rows = rng.integers(0, N, per)
        Qp = X[torch.from_numpy(rows).cuda()].double().cpu().numpy() + 0.02 * rng.standard_normal((per, D)) / np.sqrt(D)
        Qi = C[rng.integers(0, 1024, per)] + 0.5 * rng.standard_normal((per, D))
        for q in np.concatenate([Qp, Qi]):
            q = np.ascontiguousarray(q / np.linalg.norm(q))
            try:
                want, lq = ref.search(q, tau, fused=True)
            except oracle_c.ZeroLambda:
                zero += 1
                with pytest.raises(asp.PanicException):
                    aspace.search(q, gl, tau)
                continue
            got = aspace.search(q, gl, tau)
            operands.add(aspace.last_scan_operand)
            assert_hits_match(got, want, ref.scores(q, tau, lq), rtol=1e-9)
            assert abs(aspace.query_lambda(q, gl) - lq) <= 1e-9 * abs(lq)
    assert "int8-high" in operands                      # the coarse scan did serve these queries
    assert zero < per * 3                                # (some in-distribution draws have no neighbour inside eps; not all)
    c = aspace.search_counters()
    assert c["searches_with_rerun"] <= 0.05 * c["searches"]


def test_shared_scans_at_headline_size(big):
    """Concurrent callers at 1M x 768: their scans are shared (scan_tile_gang_kernel, chunks handed out by tickets at this size --
    the dynamic schedule needs three chunks per wave: nothing smaller exercises it), every call's hits are the serial ones
    (Python threads: element for element; native threads against the C ABI: the best hit of every call)."""
    import threading

    import torch

    import pyarrowspace_amd as asp
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import thread_bench
    X, aspace, gl = big["X"], big["aspace"], big["gl"]
    rng = np.random.default_rng(5)
    rows = rng.integers(0, N, 96)
    Q = X[torch.from_numpy(rows).cuda()].double().cpu().numpy() + 0.02 * rng.standard_normal((96, D)) / np.sqrt(D)
    Q = np.ascontiguousarray(Q / np.linalg.norm(Q, axis=1, keepdims=True))
    want, keep = [], []
    for i, q in enumerate(Q):
        try:
            want.append(aspace.search(q, gl, 0.62))
            keep.append(i)
        except asp.PanicException:
            pass
    Q = np.ascontiguousarray(Q[keep])
    assert len(Q) >= 48
    before = aspace.gang_counters()
    bad = []

    def worker(t):
        for i in range(60):
            j = (t * 29 + i) % len(Q)
            if aspace.search(Q[j], gl, 0.62) != want[j]:
                bad.append((t, i))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not bad, bad[:5]
    first = np.array([w[0][0] for w in want], dtype=np.int64)
    for nthr in (2, 4):
        rate, errs, gangs = thread_bench.native_rate(aspace, gl, Q, 0.62, nthr, 120, first)
        assert errs == 0 and rate > 0
    after = aspace.gang_counters()
    assert sum(after[1:]) > sum(before[1:]), (before, after)      # scans with two or more members did form
