"""GPU: k above 56 (GRAPH_VARIABLES.md:17 -- "any positive integer up to N - 1"; this build: up to 120).  The k-NN
candidate lists are 128 wide there (two slots per lane where one wave owns a list): index build, single and batched
search, the fallback paths (fp64, wavefront lists of 128 slots), the staged path, the ring build over blocks, a
duplicate-heavy set (band pass) -- against the oracle, as tests/test_gpu_parity.py does for k <= 56."""
import numpy as np
import pytest

from conftest import assert_hits_match, calibrate_eps, clustered
from test_gpu_parity import RTOL, _build_both, _check_index

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,d,k,topk,metric,kernel", [(1200, 48, 57, 10, "l2", "gaussian"), (3000, 96, 100, 15, "cosine", "rational"),
                                                       (2000, 768, 120, 40, "l2", "gaussian"), (130, 16, 120, 5, "l2", "rational")])
def test_build_and_search_match_oracle_at_wide_k(oracle_lib, n, d, k, topk, metric, kernel):
    import pyarrowspace_amd as asp
    X = clustered(n, d, nclust=4, seed=n + k)
    gp = {"eps": calibrate_eps(X, min(k, n // 3), metric), "k": k, "topk": topk, "p": 2.0, "sigma": None, "metric": metric, "kernel": kernel}
    aspace, gl, ref = _build_both(X, gp, oracle_lib)
    _check_index(aspace, gl, ref)
    assert ref.knn_cnt.max() > 56              # the lists really are wider than one slot per lane
    rng = np.random.default_rng(7)
    Q = np.stack([X[rng.integers(0, n)] + 0.05 * rng.standard_normal(d) / np.sqrt(d) for _ in range(5)])
    for q in Q:
        for tau in (1.0, 0.62, 0.3, 0.0):
            want, lq = ref.search(q, tau)
            assert_hits_match(aspace.search(np.ascontiguousarray(q), gl, tau), want, ref.scores(q, tau, lq), rtol=RTOL)
            assert abs(aspace.query_lambda(np.ascontiguousarray(q), gl) - lq) <= RTOL * abs(lq)
    if d <= 1024:
        got = aspace.search_batch(Q, gl, 0.62)
        assert got == [aspace.search(np.ascontiguousarray(q), gl, 0.62) for q in Q]
    with pytest.raises(ValueError, match="maximum of 120"):
        asp.ArrowSpaceBuilder.build(dict(gp, k=121), clustered(400, 8, nclust=2, seed=1))


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_fallback_paths_at_wide_k(oracle_lib, mode):
    """bit0 = fp64 end to end, bit1 = wavefront lists (128 slots: WaveList2) instead of the candidate buffers."""
    import pyarrowspace_amd as asp
    n, d, k = 2500, 64, 90
    X = clustered(n, d, nclust=5, seed=21)
    gp = {"eps": calibrate_eps(X, k), "k": k, "topk": 12, "p": 2.0, "sigma": None, "_search_mode": mode}
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    ref = oracle_lib.OracleIndex(X, gp)
    np.testing.assert_allclose(aspace.lambdas(), ref.lambdas, rtol=RTOL)
    rng = np.random.default_rng(3)
    for _ in range(4):
        q = X[rng.integers(0, n)] + 0.02 * rng.standard_normal(d) / np.sqrt(d)
        for tau in (1.0, 0.62, 0.0):
            want, lq = ref.search(q, tau)
            assert_hits_match(aspace.search(q, gl, tau), want, ref.scores(q, tau, lq), rtol=RTOL)


def test_exact_build_fallback_and_duplicates_at_wide_k(oracle_lib):
    """force_exact: every row through the row-serial fp64 path (wavefront lists of 128 slots); and a set with 150 copies
    of one item: more exact ties than a list holds at the k-th distance -> band pass."""
    n, d, k = 500, 24, 80
    X = clustered(n, d, nclust=3, seed=5)
    gp = {"eps": calibrate_eps(X, k), "k": k, "topk": 5, "p": 2.0, "sigma": None, "force_exact": True}
    aspace, gl, ref = _build_both(X, gp, oracle_lib)
    _check_index(aspace, gl, ref)
    n = 3000
    X = clustered(n, d, nclust=3, seed=6, normalise=False)
    X[np.random.default_rng(1).choice(n, 150, replace=False)] = X[7]
    gp = {"eps": calibrate_eps(clustered(n, d, nclust=3, seed=6, normalise=False), k), "k": k, "topk": 5, "p": 2.0, "sigma": None}
    aspace, gl, ref = _build_both(X, gp, oracle_lib)
    _check_index(aspace, gl, ref)
    want, lq = ref.search(X[7].copy(), 0.62)
    assert_hits_match(aspace.search(X[7].copy(), gl, 0.62), want, ref.scores(X[7].copy(), 0.62, lq), rtol=RTOL)


def test_staged_and_ring_at_wide_k(oracle_lib):
    """The staged C ABI on one rank, and the ring build over three uneven blocks in one process (tests/test_gpu_ring.py's
    harness), with 128-wide lists: lists, lambdas and searches as the oracle's."""
    import torch
    from pyarrowspace_amd.dist import ShardedIndex
    n, d, k = 2400, 40, 100
    X = clustered(n, d, nclust=4, seed=31)
    gp = {"eps": calibrate_eps(X, k), "k": k, "topk": 9, "p": 2.0, "sigma": None}
    ref = oracle_lib.OracleIndex(X, gp)
    index = ShardedIndex.build(gp, torch.from_numpy(X).cuda())
    np.testing.assert_allclose(index.lambdas(), ref.lambdas, rtol=RTOL)
    rng = np.random.default_rng(5)
    for _ in range(4):
        q = X[rng.integers(0, n)] + 0.02 * rng.standard_normal(d) / np.sqrt(d)
        for tau in (1.0, 0.62):
            want, lq = ref.search(q, tau)
            assert_hits_match(index.search(q, tau), want, ref.scores(q, tau, lq), rtol=RTOL)
            assert abs(index.last_lambda_q - lq) <= RTOL * abs(lq)
    index.close()
    from test_gpu_ring import _symmetric_ring_lists
    cuts = [0, 700, 1500, n]
    res, _ = _symmetric_ring_lists(X, gp, cuts)
    for b in range(len(cuts) - 1):
        lo, hi = cuts[b], cuts[b + 1]
        idx, _dist, _gy, cnt = res[b]
        np.testing.assert_array_equal(cnt, ref.knn_cnt[lo:hi])
        for r in range(hi - lo):
            assert list(idx[r, : cnt[r]]) == list(ref.knn_idx[lo + r, : cnt[r]]), (b, r)


def test_ring_second_and_third_round_at_wide_k(oracle_lib):
    """k = 100 on the ring with 5 000 exact copies of one item inside one block and 500 spread over all blocks: the second
    round (band per block) and the third (exact evaluation of every pair: as_knn_block_exact) run with 128-wide lists --
    lists equal to the oracle's (tools/fuzz_2rank.py found the third round refusing k above 64)."""
    from test_gpu_ring import _ring_lists, _symmetric_ring_lists
    rng = np.random.default_rng(11)
    n, d, k = 9000, 24, 100
    X = clustered(n, d, nclust=10, seed=23, normalise=False)
    item = X[7].copy()
    X[2100:7100] = item
    X[rng.choice(n, 500, replace=False)] = item
    gp = {"eps": calibrate_eps(clustered(n, d, nclust=10, seed=23, normalise=False), k), "k": k, "topk": 5, "p": 2.0, "sigma": None}
    ref = oracle_lib.OracleIndex(X, gp)
    cuts = [0, 2000, 7500, 9000]
    for b in range(3):
        lo, hi = cuts[b], cuts[b + 1]
        idx, key, dist, gy, cnt, nflag, over = _ring_lists(X, gp, cuts, b)
        assert nflag > 0 and over > 0
        np.testing.assert_array_equal(cnt, ref.knn_cnt[lo:hi])
        np.testing.assert_array_equal(idx, ref.knn_idx[lo:hi])
    res, flagged = _symmetric_ring_lists(X, gp, cuts)
    assert flagged > 0
    for b in range(3):
        lo, hi = cuts[b], cuts[b + 1]
        np.testing.assert_array_equal(res[b][3], ref.knn_cnt[lo:hi])
        np.testing.assert_array_equal(res[b][0], ref.knn_idx[lo:hi])


def test_wide_k_at_200k_items_properties():
    """200 000 x 256, k = 100, topk = 50 (lists of 128, symmetric pass, 3 x the headline's list traffic) -- past what the
    oracle finishes in seconds, so by properties (tests/test_gpu_fullsize.py): Laplacian identities, sampled rows' k-NN
    against an independent fp64 brute force (torch), scores recomputed from the accessors, no left-out item beats the
    last hit, batched == single."""
    import torch

    import pyarrowspace_amd as asp
    from conftest import gpu_clustered
    from test_gpu_fullsize import check_laplacian, check_sampled_knn, check_search
    n, d, k, topk = 200_000, 256, 100, 50
    X = gpu_clustered(n, d, 11, nclust=512)
    import bench
    eps = bench.calibrate_eps(X, k)
    gp = {"eps": eps, "k": k, "topk": topk, "p": 2.0, "sigma": None}
    aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, d, d)
    st = gl.build_stats()
    assert st["unproven_rows"] == 0
    csr = gl.to_csr()
    assert int(np.diff(csr[0]).max()) - 1 > 56          # rows with more than 56 neighbours exist: the wide lists were needed
    check_laplacian(csr, gl.degrees(), n, k)
    check_sampled_knn(X, csr, "l2", eps, k, nsample=48)
    lam = np.asarray(aspace.lambdas())
    rows = np.random.default_rng(3).integers(0, n, 5)
    for tau in (1.0, 0.62):
        check_search(X, aspace, gl, lam, tau, topk, rows)
    Q = np.stack([X[int(i)].double().cpu().numpy() * 1.01 for i in rows])
    assert aspace.search_batch(Q, gl, 0.62) == [aspace.search(q, gl, 0.62) for q in Q]
    del aspace, gl
    torch.cuda.empty_cache()


def test_eight_ranks_worth_of_records_at_k_120(oracle_lib):
    """k = 120 on 8 ranks is 960 neighbour records per query (the merge takes 1 024): the staged steps of the C ABI driven
    for eight row ranges of one space in one process -- every range's scan leaves its k records, lambda_q is formed from
    all of them, every range scores against it, the hit records are merged -- against the oracle."""
    import torch
    from pyarrowspace_amd.dist import ShardedIndex, shard_bounds
    n, d, k, topk, G = 4000, 48, 120, 12, 8
    X = clustered(n, d, nclust=3, seed=77)
    gp = {"eps": calibrate_eps(X, k), "k": k, "topk": topk, "p": 2.0, "sigma": None}
    ref = oracle_lib.OracleIndex(X, gp)
    index = ShardedIndex.build(gp, torch.from_numpy(X).cuda())      # one rank holds every row; the ranges below play eight
    e = index.engine
    b = shard_bounds(n, G)
    rng = np.random.default_rng(9)
    try:
        for _ in range(3):
            q = np.ascontiguousarray(X[rng.integers(0, n)] + 0.02 * rng.standard_normal(d) / np.sqrt(d))
            for tau in (1.0, 0.62):
                e.set_mode(0)
                recs = []
                for r in range(G):
                    e.query_scan(q, b[r], b[r + 1])
                    torch.cuda.synchronize()
                    recs.append(e.knn_local.clone())
                knn_all = torch.cat(recs).contiguous()
                assert knn_all.shape[0] == G * k == 960
                hits = []
                for r in range(G):
                    e.query_scan(q, b[r], b[r + 1])
                    e.query_lambda(knn_all)
                    e.query_score(tau)
                    torch.cuda.synchronize()
                    hits.append(e.hits_local.clone())
                got, lq, zero, inexact, overflow = e.query_finish(torch.cat(hits).contiguous())
                assert not zero and not inexact and not overflow
                want, lq_ref = ref.search(q, tau)
                assert abs(lq - lq_ref) <= RTOL * abs(lq_ref)
                assert_hits_match(got, want, ref.scores(q, tau, lq_ref), rtol=RTOL)
    finally:
        index.close()
