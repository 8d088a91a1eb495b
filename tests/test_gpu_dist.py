"""GPU: the staged C ABI (what multi-GPU hosts call) through pyarrowspace_amd.dist.HipEngine.
world_size 1 in-process, and world_size 2 with both ranks sharing the one visible GPU
(records staged through the CPU because RCCL wants one GPU per rank; on a multi-GPU node
bench.py --gpus N uses the nccl backend directly)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import assert_hits_match, calibrate_eps, clustered  # noqa: E402
from pyarrowspace_amd import PanicException  # noqa: E402

pytestmark = pytest.mark.gpu


def _queries(X, n, d):
    rng = np.random.default_rng(5)
    return [(X[rng.integers(0, n)] + 0.02 * rng.standard_normal(d) / np.sqrt(d), tau)
            for _ in range(4) for tau in (1.0, 0.62, 0.0)]


def test_single_rank_staged_path_matches_oracle(oracle_lib):
    import torch
    from pyarrowspace_amd.dist import ShardedIndex
    n, d = 1200, 64
    X = clustered(n, d, nclust=8, seed=31)
    gp = {"eps": calibrate_eps(X, 8), "k": 8, "topk": 6, "p": 2.0, "sigma": None}
    index = ShardedIndex.build(gp, torch.from_numpy(X).cuda())
    ref = oracle_lib.OracleIndex(X, gp)
    np.testing.assert_allclose(index.lambdas(), ref.lambdas, rtol=1e-9)
    for q, tau in _queries(X, n, d):
        want, lq = ref.search(q, tau)
        got = index.search(q, tau)
        assert_hits_match(got, want, ref.scores(q, tau, lq), rtol=1e-9)   # ties to rounding (tau = 0) may swap
        assert abs(index.last_lambda_q - lq) <= 1e-9 * abs(lq)
    Qb = np.stack([q for q, _ in _queries(X, n, d)] * 4)               # 48 queries: two passes of 32 slots
    for tau in (0.62, 0.0):
        got = index.search_batch(Qb, tau)
        for b in range(len(Qb)):
            want, lq = ref.search(Qb[b], tau)
            assert_hits_match(got[b], want, ref.scores(Qb[b], tau, lq), rtol=1e-9)
    index.close()


def test_staged_path_repairs_a_crowded_neighbourhood(oracle_lib):
    """eps above the diameter: the scan's candidate buffer (4096) overflows, the staged host asks for the
    threshold repair over the kept dots (set_exact bit2) instead of a second scan."""
    import torch
    from pyarrowspace_amd import dist as asdist
    n, d = 5000, 64
    X = clustered(n, d, nclust=6, noise=0.4, seed=23)
    gp = {"eps": 10.0, "k": 25, "topk": 15, "p": 2.0, "sigma": None}
    index = asdist.ShardedIndex.build(gp, torch.from_numpy(X).cuda())
    ref = oracle_lib.OracleIndex(X, gp)
    modes = []
    real = index.engine.set_mode
    index.engine.set_mode = lambda m: (modes.append(m), real(m))[1]
    for q, tau in _queries(X, n, d)[:6]:
        want, lq = ref.search(q, tau)
        got = index.search(q, tau)
        assert_hits_match(got, want, ref.scores(q, tau, lq), rtol=1e-9)
        assert abs(index.last_lambda_q - lq) <= 1e-9 * abs(lq)
    assert 4 in modes and 2 not in modes and 6 not in modes, modes
    index.close()


def _cpu_staged():
    """ShardedIndex with every collective staged through the CPU (gloo): two ranks can then share ONE GPU."""
    from pyarrowspace_amd.dist import HostStagedIndex
    return HostStagedIndex


def _worker(rank, world, port, n, d, split, out, replicate=False):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        CpuStaged = _cpu_staged()

        X = clustered(n, d, nclust=8, seed=31)
        gp = {"eps": calibrate_eps(X, 8), "k": 8, "topk": 6, "p": 2.0, "sigma": None}
        bounds = [0, split, n]
        shard = torch.from_numpy(X[bounds[rank] : bounds[rank + 1]].copy()).cuda()
        index = CpuStaged.build(gp, shard, dist, replicate=replicate)
        assert index.replicated == replicate and getattr(index, "ring_symmetric", False) == (not replicate)
        # the ring's block passes ran on the shards' int8 images (clustered rows: every rank's image allows it; the ranks
        # agreed through one all-gather of their measured maxima) -- the lists below are those of the fp64 oracle all the same
        assert replicate or index.ring_i8
        res = [(index.search(q, tau), index.last_lambda_q) for q, tau in _queries(X, n, d)]
        rng = np.random.default_rng(6)
        Qb = np.stack([X[rng.integers(0, n)] + 0.02 * rng.standard_normal(d) / np.sqrt(d) for _ in range(45)])
        for tau in (0.62, 1.0):   # two passes of 32 slots, the second partly filled
            assert index.search_batch(Qb, tau) == [index.search(np.ascontiguousarray(q), tau) for q in Qb]
        # sharded save / load: one file per rank, the loaded index answers like the one that was saved
        prefix = os.path.join(os.environ.get("TMPDIR", "/tmp"), "as_shard_%d_%d" % (port, int(replicate)))
        index.save(prefix)
        dist.barrier()
        loaded = CpuStaged.load(prefix, gp, dist)
        assert (loaded.n, loaded.replicated) == (index.n, index.replicated)
        assert replicate or (loaded.r0, loaded.r1) == (index.r0, index.r1)   # a replicated index is re-partitioned evenly
        for q, tau in _queries(X, n, d)[:4]:
            assert loaded.search(q, tau) == index.search(q, tau)
        loaded.close()
        os.remove("%s.rank%dof%d" % (prefix, rank, world))
        lists = None
        if not replicate:   # the rank's k-NN lists (global ids), for a diagnosis by row when something is off
            lists = tuple(t.cpu().numpy().copy() for t in index.engine.lists())
        out[rank] = (index.lambdas().copy(), res, lists, (bounds[rank], bounds[rank + 1]))
        index.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("replicate", [False, True], ids=["ring", "replicated"])
def test_two_ranks_one_gpu_match_oracle(oracle_lib, replicate):
    """ring: each rank ingests only its rows, the other shard visits by send/recv; replicated: round-1 form."""
    import torch.multiprocessing as mp
    n, d, world, split = 1200, 64, 2, 500
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, n, d, split, out, replicate), nprocs=world, join=True)
    X = clustered(n, d, nclust=8, seed=31)
    gp = {"eps": calibrate_eps(X, 8), "k": 8, "topk": 6, "p": 2.0, "sigma": None}
    ref = oracle_lib.OracleIndex(X, gp)
    want = [ref.search(q, tau) for q, tau in _queries(X, n, d)]
    for rank in range(world):
        lam, res, lists, (lo, hi) = out[rank]
        if lists is not None:   # first the lists, row by row: a wrong lambda says nothing about where it came from
            idx, dist_, gy, cnt = lists
            bad = [r for r in range(hi - lo) if cnt[r] != ref.knn_cnt[lo + r] or list(idx[r, : cnt[r]]) != list(ref.knn_idx[lo + r, : cnt[r]])]
            assert not bad, "rank %d: %d rows with wrong lists, first %s: got %s want %s (nan dists: %d)" % (
                rank, len(bad), bad[:8], idx[bad[0]], ref.knn_idx[lo + bad[0]], int(np.isnan(dist_).sum()))
        np.testing.assert_allclose(lam, ref.lambdas, rtol=1e-9)
        for (hits, lq), (whits, wlq) in zip(res, want):
            assert_hits_match(hits, whits, rtol=1e-9)   # ties to rounding (tau = 0) may swap
            assert abs(lq - wlq) <= 1e-9 * abs(wlq)
    assert out[0][1] == out[1][1]


def _dup_worker(rank, world, port, n, d, split, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, gp = _dup_data(n, d)
        bounds = [0, split, n]
        index = _cpu_staged().build(gp, torch.from_numpy(X[bounds[rank] : bounds[rank + 1]].copy()).cuda(), dist)
        lists = tuple(t.cpu().numpy().copy() for t in index.engine.lists())
        out[rank] = (index.lambdas().copy(), lists, index.ring_flagged, index.ring_overflowed, index.search(X[5000].copy(), 0.62))
        index.close()
    finally:
        dist.destroy_process_group()


def _dup_data(n, d):
    """6 000 exact copies of one item in rows 3000..8999 and 60 more among the first 2 000 rows."""
    X = clustered(n, d, nclust=8, seed=37, normalise=False)
    item = X[11].copy()
    X[3000:9000] = item
    X[np.random.default_rng(3).choice(2000, 60, replace=False)] = item
    gp = {"eps": calibrate_eps(clustered(n, d, nclust=8, seed=37, normalise=False), 8), "k": 8, "topk": 6, "p": 2.0, "sigma": None}
    return X, gp


def test_two_ranks_third_round_is_collective(oracle_lib):
    """Rank 1's block holds 6 000 copies of one item: the band of every copy (and of rank 0's 60 copies) overflows the
    collection buffers there, so both ranks go round a third time (exact evaluation); lists and lambdas as the oracle's."""
    import torch.multiprocessing as mp
    n, d, world, split = 12000, 16, 2, 2000
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dup_worker, args=(world, port, n, d, split, out), nprocs=world, join=True)
    X, gp = _dup_data(n, d)
    ref = oracle_lib.OracleIndex(X, gp)
    bounds = [0, split, n]
    for rank in range(world):
        lam, (idx, dist_, gy, cnt), flagged, over, hits = out[rank]
        lo, hi = bounds[rank], bounds[rank + 1]
        assert flagged > 0 and over > 0, (rank, flagged, over)
        np.testing.assert_array_equal(cnt, ref.knn_cnt[lo:hi])
        np.testing.assert_array_equal(idx, ref.knn_idx[lo:hi])
        np.testing.assert_allclose(lam, ref.lambdas, rtol=1e-9)
    assert out[0][4] == out[1][4]
    assert_hits_match(out[0][4], ref.search(X[5000].copy(), 0.62)[0], rtol=1e-9)


_TINY = [   # (n, d, split, graph_params): shards of a few rows, topk beyond a shard's (or the index's) rows
    (5, 2, 2, {"eps": 0.9, "k": 2, "topk": 19, "p": 0.5, "sigma": None, "metric": "cosine", "kernel": "gaussian"}),
    (9, 3, 8, {"eps": 6.0, "k": 3, "topk": 7, "p": 2.0, "sigma": None}),
    (40, 8, 1, {"eps": 8.0, "k": 5, "topk": 30, "p": 2.0, "sigma": None}),
]


def _tiny_data(n, d, seed):
    return np.abs(np.random.default_rng(seed).standard_normal((n, d))) + 0.05


def _tiny_worker(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        res = []
        for c, (n, d, split, gp) in enumerate(_TINY):
            X = _tiny_data(n, d, c)
            bounds = [0, split, n]
            index = _cpu_staged().build(gp, torch.from_numpy(X[bounds[rank] : bounds[rank + 1]].copy()).cuda(), dist)
            qs = [X[0].copy(), X[n - 1] * 1.01, X[n // 2] + 0.01]
            hits = [index.search(np.ascontiguousarray(q), tau) for q in qs for tau in (1.0, 0.62)]
            assert index.search_batch(np.stack(qs), 0.62) == hits[1::2]
            res.append((index.lambdas().copy(), hits))
            index.close()
        out[rank] = res
    finally:
        dist.destroy_process_group()


def test_two_ranks_with_shards_of_a_few_rows(oracle_lib):
    """topk larger than a rank's row count (19 of 2 + 3 rows), a one-row shard: the hit records every rank contributes
    have one layout -- capped by the items of the whole index, not by the shard (found by tools/fuzz_2rank.py)."""
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_tiny_worker, args=(2, port, out), nprocs=2, join=True)
    assert out[0][0][1] == out[1][0][1]
    for c, (n, d, split, gp) in enumerate(_TINY):
        X = _tiny_data(n, d, c)
        ref = oracle_lib.OracleIndex(X, gp)
        lam, hits = out[0][c]
        np.testing.assert_allclose(lam, ref.lambdas, rtol=1e-9)
        qs = [X[0].copy(), X[n - 1] * 1.01, X[n // 2] + 0.01]
        t = 0
        for q in qs:
            for tau in (1.0, 0.62):
                want, lq = ref.search(np.ascontiguousarray(q), tau)
                assert len(hits[t]) == min(gp["topk"], n)
                assert_hits_match(hits[t], want, ref.scores(np.ascontiguousarray(q), tau, lq), rtol=1e-9)
                t += 1


@pytest.mark.parametrize("rep", range(4))
def test_two_ranks_ring_repeats(oracle_lib, rep):
    """The 2-rank ring build again and again (its kernels run concurrently with the other rank's on one GPU: timing varies)."""
    test_two_ranks_one_gpu_match_oracle(oracle_lib, False)


def _feature_worker(rank, world, port, n, d, split, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from conftest import calibrate_feature_eps
        from pyarrowspace_amd.dist import HostStagedIndex as CpuStaged

        X = clustered(n, d, nclust=8, seed=31)
        gp = {"eps": calibrate_feature_eps(X, 8), "k": 8, "topk": 6, "p": 2.0, "sigma": None, "metric": "cosine", "kernel": "rational",
              "lambda_mode": "feature"}
        bounds = [0, split, n]
        index = CpuStaged.build(gp, torch.from_numpy(X[bounds[rank] : bounds[rank + 1]].copy()).cuda(), dist)
        res = [(index.search(q, tau), index.last_lambda_q) for q, tau in _queries(X, n, d)]
        Qb = np.stack([q for q, _ in _queries(X, n, d)])
        assert index.search_batch(Qb, 0.62) == [index.search(np.ascontiguousarray(q), 0.62) for q in Qb]
        # sharded save / load in feature mode: the rank's rows + the feature graph; the loaded index knows how many items
        # the lambdas were ranked over (all ranks' rows) and answers like the one that was saved
        prefix = os.path.join(os.environ.get("TMPDIR", "/tmp"), "as_fshard_%d" % port)
        index.save(prefix)
        dist.barrier()
        loaded = CpuStaged.load(prefix, gp, dist)
        assert (loaded.n, loaded.replicated, loaded.r0, loaded.r1) == (index.n, False, index.r0, index.r1)
        for q, tau in _queries(X, n, d)[:4]:
            assert loaded.search(q, tau) == index.search(q, tau)
        # load -> save -> load: a loaded shard holds its own rows' energies only (a built one every item's) and must
        # write them from the right offset on every rank
        loaded.save(prefix + "_again")
        dist.barrier()
        again = CpuStaged.load(prefix + "_again", gp, dist)
        np.testing.assert_array_equal(again.lambdas(), index.lambdas())
        for q, tau in _queries(X, n, d)[:4]:
            assert again.search(q, tau) == index.search(q, tau)
        a, b = ("%s.rank%dof%d" % (prefix, rank, world)), ("%s_again.rank%dof%d" % (prefix, rank, world))
        assert open(a, "rb").read() == open(b, "rb").read()
        again.close()
        loaded.close()
        os.remove(a)
        os.remove(b)
        out[rank] = (index.lambdas().copy(), res)
        index.close()
    finally:
        dist.destroy_process_group()


def test_two_ranks_feature_mode_match_oracle(oracle_lib):
    """lambda_mode='feature' row-sharded: Gram partials all-gathered and summed, energies local, tau0 global."""
    import torch.multiprocessing as mp
    from conftest import calibrate_feature_eps
    n, d, world, split = 1500, 96, 2, 400
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_feature_worker, args=(world, port, n, d, split, out), nprocs=world, join=True)
    X = clustered(n, d, nclust=8, seed=31)
    gp = {"eps": calibrate_feature_eps(X, 8), "k": 8, "topk": 6, "p": 2.0, "sigma": None, "metric": "cosine", "kernel": "rational",
          "lambda_mode": "feature"}
    ref = oracle_lib.OracleIndex(X, gp)
    want = [ref.search(q, tau) for q, tau in _queries(X, n, d)]
    for rank in range(world):
        lam, res = out[rank]
        np.testing.assert_allclose(lam, ref.lambdas, rtol=1e-9)
        for (hits, lq), (whits, wlq) in zip(res, want):
            assert_hits_match(hits, whits, rtol=1e-9)
            assert abs(lq - wlq) <= 1e-9 * abs(wlq)
    assert out[0][1] == out[1][1]


@pytest.mark.parametrize("library_exchange", [True, False], ids=["library-rccl", "torch-collectives"])
def test_one_rank_rccl_collectives_on_a_side_stream(oracle_lib, library_exchange, monkeypatch):
    """The real N>1 code path with a 1-rank nccl group: every collective is issued, nothing is skipped.
    library-rccl: the two per-query all-gathers are issued by the library itself (as_query_search_staged: RCCL from C++
    on the query's stream, one host call per query); torch-collectives: torch.distributed's all_gather_into_tensor
    ordered against the query kernels on one dedicated stream (ARROWSPACE_PY_COLLECTIVES=1)."""
    import torch
    import torch.distributed as dist
    from pyarrowspace_amd.dist import ShardedIndex
    if library_exchange:
        monkeypatch.delenv("ARROWSPACE_PY_COLLECTIVES", raising=False)
    else:
        monkeypatch.setenv("ARROWSPACE_PY_COLLECTIVES", "1")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n, d = 5000, 128
        X = clustered(n, d, nclust=16, seed=41)
        gp = {"eps": calibrate_eps(X, 10), "k": 10, "topk": 8, "p": 2.0, "sigma": None}
        index = ShardedIndex.build(gp, torch.from_numpy(X).cuda(), dist, force_collectives=True)
        assert index._lib_comm == library_exchange
        ref = oracle_lib.OracleIndex(X, gp)
        np.testing.assert_allclose(index.lambdas(), ref.lambdas, rtol=1e-9)
        for rep in range(3):                      # repeated: a missing stream dependency shows up as stale records
            for q, tau in _queries(X, n, d):
                want, lq = ref.search(q, tau)
                assert_hits_match(index.search(q, tau), want, ref.scores(q, tau, lq))
                assert abs(index.last_lambda_q - lq) <= 1e-9 * abs(lq)
        far = np.zeros(d)
        far[0] = 50.0
        with pytest.raises(PanicException):       # no neighbour within eps: the reference's zero-lambda assert, on this path too
            index.search(far, 0.62)
        Qb = np.stack([q for q, _ in _queries(X, n, d)])
        assert index.search_batch(Qb, 0.62) == [index.search(np.ascontiguousarray(q), 0.62) for q in Qb]
        index.close()
    finally:
        dist.destroy_process_group()


def test_library_exchange_repairs_a_crowded_neighbourhood(oracle_lib):
    """eps above the diameter on the library-side exchange path: the escalation of as_query_search_staged (threshold
    repair over the kept dots, mode 4) is the Python host's (next_mode), step for step."""
    import torch
    import torch.distributed as dist
    from pyarrowspace_amd.dist import ShardedIndex
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        n, d = 5000, 64
        X = clustered(n, d, nclust=6, noise=0.4, seed=23)
        gp = {"eps": 10.0, "k": 25, "topk": 15, "p": 2.0, "sigma": None}
        index = ShardedIndex.build(gp, torch.from_numpy(X).cuda(), dist, force_collectives=True)
        assert index._lib_comm
        ref = oracle_lib.OracleIndex(X, gp)
        for q, tau in _queries(X, n, d)[:6]:
            want, lq = ref.search(q, tau)
            assert_hits_match(index.search(q, tau), want, ref.scores(q, tau, lq), rtol=1e-9)
            assert abs(index.last_lambda_q - lq) <= 1e-9 * abs(lq)
        index.close()
    finally:
        dist.destroy_process_group()


def test_one_exchange_pass_returns_what_the_two_exchange_chain_returns(oracle_lib, monkeypatch):
    """tau in [0.4, 1]: the sharded search takes ONE exchange -- every rank's k-NN records and its scorer candidates finished to
    (id, exact cosine, lambda) in one block, lambda_q / scores / ranking redundantly behind the all-gather (as_query_x1_*).
    Same hits, bit for bit, as the two-exchange chain (the workspaces' switch off: as_query_set_x1 -- ARROWSPACE_STAGED_X1=0 when
    they are made; a row-sharded index agrees on it over its ranks, never per call) and as one space; tau below 0.4 keeps the chain."""
    import torch
    import pyarrowspace_amd as asp
    from pyarrowspace_amd.dist import ShardedIndex
    n, d = 6000, 96
    X = clustered(n, d, nclust=12, seed=77)
    gp = {"eps": calibrate_eps(X, 10), "k": 10, "topk": 8, "p": 2.0, "sigma": None}
    index = ShardedIndex.build(gp, torch.from_numpy(X).cuda())
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    ref = oracle_lib.OracleIndex(X, gp)
    rng = np.random.default_rng(9)
    Q = [np.ascontiguousarray(X[rng.integers(0, n)] + 0.03 * rng.standard_normal(d) / np.sqrt(d)) for _ in range(12)]
    for tau in (0.62, 1.0, 0.4):
        for q in Q:
            before = index.engine.x1_passes()
            index.engine.x1_set_enabled(True)
            got = index.search(q, tau)
            took = index.engine.x1_passes() - before
            assert took in (1, 2)     # (2: the coarse scan's candidates did not fit, the pass ran once more on the two-digit image)
            index.engine.x1_set_enabled(False)
            chain = index.search(q, tau)
            assert index.engine.x1_passes() == before + took
            assert got == chain == aspace.search(q, gl, tau)
            want, lq = ref.search(q, tau)
            assert_hits_match(got, want, ref.scores(q, tau, lq), rtol=1e-9)
    index.engine.x1_set_enabled(True)
    before = index.engine.x1_passes()
    assert index.search(Q[0], 0.2) == aspace.search(Q[0], gl, 0.2)
    assert index.engine.x1_passes() == before
    far = np.zeros(d)
    far[0] = 50.0
    with pytest.raises(PanicException):
        index.search(far, 0.62)
    index.close()


@pytest.mark.parametrize("world,rows,k", [(1, 700, 9), (3, 5000, 25), (8, 20000, 12), (64, 3000, 5)])
def test_edge_bucketing_kernels_match_the_torch_ops(world, rows, k, monkeypatch):
    """as_edges_bucket (count / scan / scatter kernels in front of the sharded graph stage's all-to-all) against the torch ops
    it replaces (bucketize, stable argsort, bincount, gathers): same entries in the same order, same counts -- ragged lists,
    empty rows, targets on every rank, buckets that stay empty."""
    import torch
    from pyarrowspace_amd.dist import ShardedIndex, HipEngine
    rng = np.random.default_rng(world * 1000 + rows)
    cuts = np.sort(rng.integers(0, 50000, size=world - 1)) if world > 1 else np.zeros(0, dtype=np.int64)
    if world == 8:
        cuts[3] = cuts[2]                       # an empty shard: its bucket stays empty
    bounds = [0] + [int(c) for c in cuts] + [50000]
    rank = world // 2
    idx = torch.from_numpy(rng.integers(0, 50000, size=(rows, k)).astype(np.int32)).cuda()
    cnt = torch.from_numpy(rng.integers(0, k + 1, size=rows).astype(np.int32)).cuda()
    dst = torch.from_numpy(rng.random((rows, k))).cuda()
    gy = torch.from_numpy(rng.standard_normal((rows, k))).cuda()
    index = ShardedIndex()
    index.torch, index.dist, index.group = torch, None, None
    index.world, index.rank, index.bounds, index.r0 = world, rank, bounds, bounds[rank]
    index._collective = lambda: False            # the bucketing alone: what would be handed to the all-to-all
    index.engine = HipEngine({"eps": 1.0, "k": k, "topk": 3, "p": 2.0, "sigma": None})
    monkeypatch.setenv("ARROWSPACE_TORCH_EDGES", "1")
    want = index._exchange_edges(idx, dst, gy, cnt)
    monkeypatch.delenv("ARROWSPACE_TORCH_EDGES")
    got = index._exchange_edges(idx, dst, gy, cnt)
    assert len(got) == 4
    for g, w in zip(got, want):
        assert g.dtype == w.dtype and torch.equal(g, w)
    ints, reals, send = index.engine.edge_bucket(idx, dst, gy, cnt, bounds[rank], bounds)
    assert sum(send) == int(cnt.sum().item()) and len(send) == world
    if world == 8:
        assert send[3] == 0
