"""GPU, BASELINE.json's other configurations at their full sizes, with the parameter sets of the reference's own
harnesses (SURVEY section 8d "Config restatement"); synthetic data of the stated shapes (no dataset or model is
available offline).  Same size-independent properties as tests/test_gpu_fullsize.py.
  config 2: 400k x 384, k = 4, topk = 2, sigma = .25   (/root/reference/tests/test_1_quora_questions.py:77-83),
            x100-scaled items (:74), ~1 % exact duplicate rows (the dataset is duplicate questions)
  config 3: 200k x 768, k = 25, topk = 15, sigma = None (/root/reference/tests/test_3_beir.py:194-200), x100 (:190)
  config 4: one rank's slice of 8.8M x 768 on 4 GPUs: the whole item matrix resident (27 GB fp32), exact k-NN lists
            of one 64k-row range against all 8.8M columns (as_knn_rows, what a rank computes per step).  The staged
            and batched SEARCH of that configuration is not here: tests/test_gpu_multirank.py runs it with 4 ranks at
            2M x 768, tools/config4_fullsize.py at the full 8.8M (a builder-run log, profiles/r03_config4_full.log)
The harness values of eps (0.5 and 10) are rectified-cosine distances (GRAPH_VARIABLES.md:7): they are used as
written under metric='cosine', and replaced by a calibrated eps under the north_star's L2 metric, where 0.5 / 10 on
x100-scaled items admit no edge / mean nothing."""
import ctypes as C

import numpy as np
import pytest

from conftest import brute_keys, gpu_clustered
from test_gpu_fullsize import check_laplacian, check_sampled_knn, check_search

pytestmark = pytest.mark.gpu


def _with_duplicates(X, frac, seed, group=2):
    """Overwrite frac of the rows with exact copies of other rows, in groups of `group` identical rows."""
    import torch
    n = X.shape[0]
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    ndup = int(n * frac)
    perm = torch.randperm(n, generator=g)
    dst = perm[:ndup]
    src = perm[ndup:ndup + (ndup + group - 2) // (group - 1)].repeat_interleave(group - 1)[:ndup]
    X[dst.cuda()] = X[src.cuda()]
    return dst.numpy(), src.numpy()


@pytest.mark.parametrize("metric,kernel,eps,group", [("cosine", "rational", 0.5, 2), ("l2", "gaussian", None, 2), ("l2", "gaussian", None, 40)],
                         ids=["cosine-eps0.5-pairs", "l2-calibrated-pairs", "l2-calibrated-groups-of-40"])
def test_config2_quora_shape_with_duplicates(metric, kernel, eps, group):
    import torch

    import bench
    import pyarrowspace_amd as asp
    n, d, k, topk = 400_000, 384, 4, 2
    X = gpu_clustered(n, d, 7, scale=100.0)
    dst, src = _with_duplicates(X, 0.01, 3, group)
    if eps is None:
        eps = bench.calibrate_eps(X, k, metric)
    sigma = 0.25 if metric == "cosine" else 0.5 * eps      # the harness's sigma = eps / 2
    gp = {"eps": eps, "k": k, "topk": topk, "p": 2.0, "sigma": sigma, "metric": metric, "kernel": kernel}
    aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, d, d)
    stats = gl.build_stats()
    csr = gl.to_csr()
    check_laplacian(csr, gl.degrees(), n, k)
    check_sampled_knn(X, csr, metric, eps, k, nsample=48)
    # duplicates: distance exactly 0 -- each copy and its source are mutual neighbours unless k lower-indexed
    # copies of the same vector come first (order (key, index))
    indptr, indices, _ = csr
    members = {}
    for a, b in zip(dst.tolist(), src.tolist()):
        members.setdefault(b, {b}).add(a)
    checked = 0
    for b, grp in list(members.items())[:300]:
        grp = sorted(grp)
        for a in grp:
            nb = set(indices[indptr[a]:indptr[a + 1]].tolist()) - {a}
            want = [x for x in grp if x != a][:k]          # its k lowest-indexed twins are its nearest items
            assert set(want) <= nb, (a, want, sorted(nb))
            checked += 1
    assert checked > 300
    assert stats["fallback_s"] < 0.5, stats              # duplicate rows must not fall back to the row-serial path
    lam = aspace.lambdas()
    assert np.isfinite(lam).all() and (lam >= 0).all()
    check_search(X, aspace, gl, lam, 0.62, topk, np.random.default_rng(5).choice(n, 3, replace=False))


@pytest.mark.parametrize("metric,kernel,eps", [("cosine", "rational", 10.0), ("l2", "gaussian", None)], ids=["cosine-eps10", "l2-calibrated"])
def test_config3_beir_shape(metric, kernel, eps):
    import bench
    import pyarrowspace_amd as asp
    n, d, k, topk = 200_000, 768, 25, 15
    X = gpu_clustered(n, d, 11, scale=100.0)
    if eps is None:
        eps = bench.calibrate_eps(X, k, metric)
    gp = {"eps": eps, "k": k, "topk": topk, "p": 2.0, "sigma": None, "metric": metric, "kernel": kernel}
    aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, d, d)
    csr = gl.to_csr()
    check_laplacian(csr, gl.degrees(), n, k)
    # eps = 10 is beyond the largest cosine distance: every pair is inside, the graph is the pure k-NN graph
    check_sampled_knn(X, csr, metric, eps if eps < 1 or metric == "l2" else float("inf"), k, nsample=48)
    lam = aspace.lambdas()
    assert np.isfinite(lam).all() and (lam >= 0).all()
    # every item lies inside the query's eps too: the crowded-neighbourhood repair of the search, at full size
    for tau in (1.0, 0.62, 0.51):                          # tests/test_4_msmarco_tau_sweep.py:18-22
        check_search(X, aspace, gl, lam, tau, topk, np.random.default_rng(int(tau * 100)).choice(n, 2, replace=False))


def test_config4_one_rank_slice_of_8_8M():
    import torch

    import bench
    import pyarrowspace_amd as asp
    from pyarrowspace_amd import _lib
    L = _lib.load()
    n, d, k, topk = 8_800_000, 768, 25, 15
    X = gpu_clustered(n, d, 13, nclust=8192)
    eps = bench.calibrate_eps(X, k, "l2", sample=256)
    gpd = {"eps": eps, "k": k, "topk": topk, "p": 2.0, "sigma": None}
    gp, op = asp._parse_graph_params(gpd)
    sp = C.c_void_p()
    assert L.as_space_create_dev(C.c_void_p(X.data_ptr()), _lib.DTYPE_F32, n, d, d, C.byref(op), C.byref(sp)) == 0, _lib.last_error()
    r0, r1 = 4_400_000, 4_400_000 + 65536
    rows = r1 - r0
    idx = torch.full((rows, k), -2, dtype=torch.int32, device="cuda")
    key = torch.zeros((rows, k), dtype=torch.float64, device="cuda")
    dist = torch.zeros_like(key)
    gy = torch.zeros_like(key)
    cnt = torch.zeros(rows, dtype=torch.int32, device="cuda")
    assert L.as_knn_rows(sp, C.byref(gp), r0, r1, C.c_void_p(idx.data_ptr()), C.c_void_p(key.data_ptr()), C.c_void_p(dist.data_ptr()),
                         C.c_void_p(gy.data_ptr()), C.c_void_p(cnt.data_ptr())) == 0, _lib.last_error()
    torch.cuda.synchronize()
    idx_h, key_h, cnt_h = idx.cpu().numpy(), key.cpu().numpy(), cnt.cpu().numpy()
    assert (cnt_h >= 0).all() and (cnt_h <= k).all() and cnt_h.mean() > 1
    sample = np.random.default_rng(1).choice(rows, 32, replace=False)
    keys = brute_keys(X, (sample + r0).tolist(), "l2")
    vals, bidx = torch.topk(keys, k + 8, dim=1, largest=False)
    vals, bidx = vals.cpu().numpy(), bidx.cpu().numpy()
    for t, lr in enumerate(sample):
        c = cnt_h[lr]
        got = idx_h[lr, :c]
        assert (idx_h[lr, c:] == -1).all() and len(set(got.tolist())) == c and (r0 + lr) not in got
        assert (np.diff(key_h[lr, :c]) >= 0).all() and (key_h[lr, :c] <= eps * eps).all()
        inside = [int(j) for v, j in zip(vals[t][:k], bidx[t][:k]) if v <= eps * eps - 1e-9 and (vals[t][k] - v) > 1e-9]
        assert set(inside) <= set(got.tolist()), (lr, set(inside) - set(got.tolist()))
        for v, j in zip(key_h[lr, :c], got):               # the exact keys, against the brute force
            assert abs(v - float(keys[t, j])) <= 1e-9
    L.as_free_space(sp)


def test_config5_block_pair_beyond_2G_elements():
    """config 5's per-rank shape is 8M x 768 per GPU: blocks of more than 2^31 floats, transposed buffers of more than
    2^32 entries.  as_knn_block_pair on an own block of 2.9M x 768 (2.2e9 floats) and a visiting block of 4.3M x 768
    (3.3e9 floats; 4.3M x 1024 transposed entries = 4.4e9), the own rows restricted to the last ~100k (the highest
    addresses of the own block) against ALL column tiles of the visiting block, once in one piece (35 GB of transposed buffers) and once in
    chunks of 6 719 tiles: the own rows' lists and the visiting items' lists against torch fp64 brute force."""
    import os

    import torch

    import bench
    from pyarrowspace_amd.dist import HipEngine
    na, nb, d, k = 2_900_000, 4_300_000, 768, 25
    X = gpu_clustered(na + nb, d, 17, nclust=8192)
    eps = bench.calibrate_eps(X, k, "l2", sample=256)
    gp = {"eps": eps, "k": k, "topk": 15, "p": 2.0, "sigma": None}
    A, B = X[:na], X[na:]
    row0 = (na - 100_000) // 128 * 128
    sample_a = (row0 + np.random.default_rng(1).choice(na - row0, 48, replace=False)).tolist()
    keys_a = brute_keys(X, sample_a, "l2")[:, na:].cpu().numpy()                  # own rows against the visiting block
    # visiting items with neighbours among the own rows: the nearest items of a few own rows
    near = torch.topk(torch.from_numpy(keys_a[:24]), 2, dim=1, largest=False)[1].reshape(-1).unique().tolist()
    sample_b = sorted(set(near) | set(np.random.default_rng(2).choice(nb, 24, replace=False).tolist()))
    keys_b = brute_keys(X, [na + j for j in sample_b], "l2")[:, row0:na].cpu().numpy()   # visiting items against the own rows
    for tiles in (1_000_000, 7_000):
        os.environ["ARROWSPACE_PAIR_CHUNK_TILES"] = str(tiles)
        try:
            ea, eb = HipEngine(gp), HipEngine(gp)
            ea.create_space(A)
            ea.ring_begin(2)
            eb.create_space(B)
            eb.ring_begin(2)
            nmax = [ea.block_nmax(ea.own_block()), eb.block_nmax(eb.own_block())]
            h = ea.open_block(B)
            P = ea.knn_block_pair(h, row0, na, -1, -1, 0, na, None, nb)
            ea.close_block(h)
            eb.fold_slice(P, nmax[0])
            del P
            fa, fb = ea.knn_merge(nmax), eb.knn_merge(nmax)
        finally:
            os.environ.pop("ARROWSPACE_PAIR_CHUNK_TILES", None)
        assert fa == 0 and fb == 0
        for e, rows, keys, goff, lo in ((ea, sample_a, keys_a, na, 0), (eb, sample_b, keys_b, row0, 0)):
            idx_h, key_h, cnt_h = (t[torch.as_tensor(rows, device="cuda")].cpu().numpy() for t in (e.l_idx, e.l_key, e.l_cnt))
            for t in range(len(rows)):
                c = int(cnt_h[t])
                got = idx_h[t, :c] - goff
                assert (idx_h[t, c:] == -1).all() and len(set(got.tolist())) == c and (got >= 0).all() and (got < keys.shape[1]).all()
                assert (np.diff(key_h[t, :c]) >= 0).all() and (key_h[t, :c] <= eps * eps).all()
                order = np.argsort(keys[t], kind="stable")[: k + 8]
                vals = keys[t][order]
                inside = [int(j) for v, j in zip(vals[:k], order[:k]) if v <= eps * eps - 1e-9 and (vals[k] - v) > 1e-9]
                assert set(inside) <= set(got.tolist()), (rows[t], set(inside) - set(got.tolist()))
                assert c >= min(k, int((keys[t] <= eps * eps - 1e-9).sum()))
                for v, j in zip(key_h[t, :c], got):
                    assert abs(v - float(keys[t, j])) <= 1e-9
        assert int(ea.l_cnt[:row0].sum()) == 0 and int(ea.l_cnt[row0:].sum()) > 0 and int(eb.l_cnt.sum()) > 0
        ea.close()
        eb.close()
        torch.cuda.empty_cache()
