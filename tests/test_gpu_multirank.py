"""GPU: BASELINE.json's two multi-GPU configurations as real multi-rank workloads -- one process per rank, every rank
holding only its rows, the whole protocol of DESIGN.md section 6 (ring of shards with every pair of shards computed
once, the split pair of an even ring, slices sent home, the edge all-to-all, the sharded graph stage, the three O(N)
all-gathers, staged single and batched search) -- on the ONE GPU of this pool: the ranks share the device, so every
exchange step is staged through host memory (pyarrowspace_amd.dist.HostStagedIndex, gloo); kernels, host logic and
results are those of an N-GPU run, the transport is not (nothing here measures scaling).

  * config 4 (MS MARCO tau sweep, 4 ranks row-sharded; tests/test_4_msmarco_tau_sweep.py:18-22: tau in {1.0, .62, .51}):
    4 processes, 2M x 768;
  * config 5 (8 ranks): the pool allows 6 processes on a card, the test runner itself being one of them once earlier
    tests have touched the GPU -- 5 processes, uneven shards (an odd ring: no split pair), 1.2M x 768.

Checked on every rank: the rank's CSR rows and lambdas are BIT-IDENTICAL to a single-space build of the same items
(rank 0 builds it, the rows are compared shard by shard), Laplacian identities over all ranks (symmetry by a random
bilinear form, L D^1/2 1 = 0), sampled rows' k-NN against an independent fp64 brute force (torch), staged search and
batched search equal on all ranks and equal to the single-space search, scores recomputed from the accessors.
Per-phase seconds of the build are printed (ShardedIndex.phase_s)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu
K, TOPK = 25, 15
TAUS = (1.0, 0.62, 0.51)


def _bounds(n, world, uneven):
    if not uneven:
        from pyarrowspace_amd.dist import shard_bounds
        return shard_bounds(n, world)
    w = np.array([1.0 + 0.35 * ((r * 7) % 5 - 2) / 2 for r in range(world)])      # shards of 65 % .. 135 % of the mean
    cuts = np.floor(np.cumsum(w) / w.sum() * n).astype(np.int64)
    cuts[-1] = n
    return [0] + [int(c) for c in cuts]


def _check_own_rows_knn(X, ip, ix, lo, metric, eps, k, rows, rel=1e-9):
    """check_sampled_knn (tests/test_gpu_fullsize.py) for rows of ONE rank's CSR (local row pointers, global column ids)."""
    import torch
    from conftest import brute_keys
    epskey = eps * eps if metric == "l2" else eps
    keys = brute_keys(X, rows, metric)
    vals, idx = torch.topk(keys, k + 8, dim=1, largest=False)
    vals, idx = vals.cpu().numpy(), idx.cpu().numpy()
    margin = rel * (float((X[:4096].double() ** 2).sum(1).max().item()) if metric == "l2" else 1.0)
    for t, i in enumerate(rows):
        cols = ix[ip[i - lo]:ip[i - lo + 1]]
        cols = set(cols[cols != i].tolist())
        inside = [int(j) for v, j in zip(vals[t][:k], idx[t][:k]) if v <= epskey - margin and (vals[t][k] - v) > margin]
        assert set(inside) <= cols, (i, set(inside) - cols)
        for v, j in zip(vals[t], idx[t]):
            assert not (v > epskey + margin and int(j) in cols), (i, int(j), v)


def _worker(rank, world, port, n, d, uneven, metric, kernel, out, single=True):
    import time

    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        import pyarrowspace_amd as asp
        from conftest import brute_keys, gpu_clustered
        from pyarrowspace_amd.dist import HostStagedIndex
        from test_gpu_fullsize import check_sampled_knn

        X = gpu_clustered(n, d, 42)
        if single:
            eps = bench.calibrate_eps(X, K, metric)
        else:   # (full size: the calibration's temporaries are 3 x the items -- one rank at a time is enough)
            box = [bench.calibrate_eps(X, K, metric) if rank == 0 else None]
            torch.cuda.empty_cache()
            dist.broadcast_object_list(box, 0)
            eps = box[0]
        gp = {"eps": eps, "k": K, "topk": TOPK, "p": 2.0, "sigma": None, "metric": metric, "kernel": kernel}
        b = _bounds(n, world, uneven)
        lo, hi = b[rank], b[rank + 1]
        shard = X[lo:hi].clone()
        if rank != 0 or not single:
            del X
            torch.cuda.empty_cache()
        dist.barrier()
        t0 = time.perf_counter()
        index = HostStagedIndex.build(gp, shard, dist)
        build_s = time.perf_counter() - t0
        assert not index.replicated and index.ring_symmetric and (index.r0, index.r1) == (lo, hi)
        ip, ix, v = index.engine.csr()
        deg = index.engine.degrees()
        lam_all = index.lambdas()

        # ---- Laplacian identities over all ranks: symmetry by a random bilinear form, L D^1/2 1 = 0
        rng = np.random.default_rng(0)
        u, w_ = rng.standard_normal(n), rng.standard_normal(n)
        rows = np.repeat(np.arange(lo, hi), np.diff(ip))
        part = torch.tensor([np.sum(v * u[rows] * w_[ix]), np.sum(v * w_[rows] * u[ix])], dtype=torch.float64)
        dist.all_reduce(part)
        assert abs(float(part[0] - part[1])) <= 1e-9 * (abs(float(part[0])) + abs(float(part[1])) + 1)
        degs = [None] * world
        dist.all_gather_object(degs, deg)
        deg_all = np.concatenate(degs)
        r = np.bincount(rows - lo, weights=v * np.sqrt(deg_all[ix]), minlength=hi - lo)
        assert np.max(np.abs(r[deg > 0])) < 1e-9
        assert (np.diff(ix)[np.diff(rows) == 0] > 0).all() and np.isfinite(lam_all).all() and (lam_all >= 0).all()

        # ---- searches: staged single + batched, tau sweep of tests/test_4_msmarco_tau_sweep.py:18-22
        qrng = np.random.default_rng(5)
        if rank == 0:
            qrows = qrng.integers(0, n, 6) if single else qrng.integers(lo, hi, 6)     # (full size: rank 0 holds its shard only)
            rowvec = (lambda i: X[int(i)]) if single else (lambda i: shard[int(i) - lo])
            Q = np.stack([rowvec(i).double().cpu().numpy() * 1.01 + 0.002 * qrng.standard_normal(d) / np.sqrt(d) for i in qrows])
            qt = torch.from_numpy(Q)
        else:
            qt = torch.empty((6, d), dtype=torch.float64)
        dist.broadcast(qt, 0)
        Q = qt.numpy()
        t1 = time.perf_counter()
        res = {tau: [index.search(np.ascontiguousarray(q), tau) for q in Q] for tau in TAUS}
        search_s = (time.perf_counter() - t1) / (len(TAUS) * len(Q))
        lqs = []
        for q in Q:
            index.search(np.ascontiguousarray(q), 0.62)
            lqs.append(index.last_lambda_q)
        for tau in TAUS:
            assert index.search_batch(Q, tau) == res[tau]
        for tau in TAUS:                      # scores from the definition (TAUMODE.md:33): cosine from the items this rank holds
            for q, hits, lq in zip(Q, res[tau], lqs):
                sc = [s for _, s in hits]
                assert len(hits) == TOPK and sc == sorted(sc, reverse=True) and len(set(j for j, _ in hits)) == TOPK
                for j, s in hits:
                    if lo <= j < hi:
                        x = shard[j - lo].double().cpu().numpy()
                        cos = float(x @ q) / np.sqrt(float(x @ x) * float(q @ q))
                        assert abs(s - (tau * cos + (1 - tau) / (1 + abs(lq - lam_all[j])))) < 1e-12

        if not single:
            # full size (tools/config4_fullsize.py): no single-space build to compare with -- sampled rows of rank 0 against an
            # independent fp64 brute force over all items (regenerated now that the ring's buffers are gone)
            if rank == 0:
                X = gpu_clustered(n, d, 42)
                _check_own_rows_knn(X, ip, ix, lo, metric, eps, K, np.random.default_rng(1).choice(np.arange(lo, hi), 32, replace=False))
                del X
            dist.barrier()
            out[rank] = dict(build_s=build_s, phases=dict(index.phase_s), stats=index.build_stats(), search_ms=search_s * 1e3,
                             flagged=getattr(index, "ring_flagged", 0), single_s=float("nan"), rows=hi - lo)
            index.close()
            return
        # ---- against ONE space holding every item (rank 0 builds it; the other ranks wait): bit-identical rows
        tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), "as_multirank_%d" % port)
        if rank == 0:
            torch.cuda.empty_cache()
            t2 = time.perf_counter()
            aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, d, d)
            single_s = time.perf_counter() - t2
            rip, rix, rv = gl.to_csr()
            check_sampled_knn(X, (rip, rix, rv), metric, eps, K, nsample=48)
            rres = {tau: [aspace.search(np.ascontiguousarray(q), gl, tau) for q in Q] for tau in TAUS}
            os.makedirs(tmp, exist_ok=True)
            for nm, arr in (("ip", rip), ("ix", rix), ("v", rv), ("lam", aspace.lambdas())):   # (files: a pickle of 1.3 GB per rank is no way to ship this)
                np.save(os.path.join(tmp, nm + ".npy"), arr)
            small = [dict(res=rres, tau0=gl.tau0, single_s=single_s)]
            del aspace, gl, rip, rix, rv
        else:
            small = [None]
        dist.broadcast_object_list(small, 0)
        small = small[0]
        rip = np.load(os.path.join(tmp, "ip.npy"), mmap_mode="r")
        a, e = int(rip[lo]), int(rip[hi])
        np.testing.assert_array_equal(ip, np.asarray(rip[lo:hi + 1]) - a)
        np.testing.assert_array_equal(ix, np.load(os.path.join(tmp, "ix.npy"), mmap_mode="r")[a:e])
        np.testing.assert_array_equal(v, np.load(os.path.join(tmp, "v.npy"), mmap_mode="r")[a:e])
        np.testing.assert_array_equal(lam_all, np.load(os.path.join(tmp, "lam.npy")))
        assert index.engine.tau0() == small["tau0"]
        for tau in TAUS:
            for got, want in zip(res[tau], small["res"][tau]):
                assert [j for j, _ in got] == [j for j, _ in want], (tau, got[:3], want[:3])
                np.testing.assert_allclose([s_ for _, s_ in got], [s_ for _, s_ in want], rtol=1e-12)
        dist.barrier()
        if rank == 0:
            import shutil
            shutil.rmtree(tmp, ignore_errors=True)
        ref = small
        out[rank] = dict(build_s=build_s, phases=dict(index.phase_s), stats=index.build_stats(), search_ms=search_s * 1e3,
                         flagged=getattr(index, "ring_flagged", 0), single_s=ref["single_s"], rows=hi - lo)
        index.close()
    finally:
        dist.destroy_process_group()


def _run(world, n, d, uneven, metric="l2", kernel="gaussian", single=True):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, n, d, uneven, metric, kernel, out, single), nprocs=world, join=True)
    assert sorted(out.keys()) == list(range(world))
    for r in range(world):
        o = out[r]
        ph = ", ".join("%s %.2f" % (k, v) for k, v in o["phases"].items())
        print("  rank %d/%d (%d rows): build %.2f s [%s]; k-NN kernel %.2f s; %d rows to the second round; staged search %.2f ms/query"
              % (r, world, o["rows"], o["build_s"], ph, o["stats"].get("knn_mfma_s", 0.0), o["flagged"], o["search_ms"]), flush=True)
    print("  single space on the same GPU: %.2f s" % out[0]["single_s"], flush=True)
    return out


def test_config4_four_ranks_2m_by_768():
    """BASELINE.json config 4 as a 4-rank job: even ring (the pair at distance 2 is split between its two ends)."""
    _run(4, 2_000_000, 768, uneven=False)


def test_config5_five_ranks_uneven_shards_cosine():
    """BASELINE.json config 5's protocol with as many ranks as the pool allows on one card next to the test runner (5),
    uneven shards, in the mode the reference's parameter sets are written for (rectified-cosine distance, rational weights)."""
    _run(5, 1_200_000, 768, uneven=True, metric="cosine", kernel="rational")


def test_config4_full_size_8_8m_by_768_four_ranks():
    """BASELINE.json config 4 at its FULL size (8.8M x 768 fp32 row-sharded over 4 ranks of 2.2M rows) as a 4-rank job on the one
    GPU of this box: the ring's block passes on the shards' int8 images, every pair of shards once, slices home, edge
    all-to-all, sharded graph stage, sharded single + batched search; verification inside the worker (Laplacian identities
    over all ranks, sampled rows against an fp64 brute force, scores from the definition, single == batched on every rank).
    About 2.5 minutes; needs most of the card's memory (skipped below 200 GB free)."""
    import torch
    free, _ = torch.cuda.mem_get_info(0)
    if free < 200e9:
        pytest.skip("needs 200 GB of free device memory: %.0f GB free" % (free / 1e9))
    out = _run(4, 8_800_000, 768, uneven=False, single=False)
    assert all(out[r]["rows"] == 2_200_000 for r in range(4))


def test_bench_line_of_a_two_rank_job():
    """`python bench.py --gpus 2` as the driver starts it (no launcher: bench.py spawns its ranks; on this one-GPU box they
    share the card and the exchange steps go through host memory): ONE JSON line from rank 0 with the N > 1 contract --
    n_gpus, ranks_seen (an all-reduce of ones over the ranks' collective layer), `roofline` a physical fraction with its
    per-rank breakdown, `cpu_baseline` on rank 0 (the oracle over ALL items, the lambdas and degrees gathered from the ranks)."""
    import json
    import subprocess

    env = dict(os.environ, ARROWSPACE_BENCH_NDEV="1", ARROWSPACE_BENCH_LAUNCH_TIMEOUT="500")
    env.pop("WORLD_SIZE", None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--n", "120000", "--d", "128", "--k", "10", "--topk", "5",
                        "--steps", "30", "--warmup", "5", "--cpu-queries", "4", "--cpu-build-n", "2000"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    out = lines[0]
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["scaling"] == "strong" and out["value"] > 0
    r = out["roofline"]
    assert r["bound"] == "hbm" and 0 < r["frac"] <= 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    pr = out["roofline_per_rank"]
    assert [e["rank"] for e in pr] == [0, 1] and sum(e["rows"] for e in pr) == 120000 and all(0 < e["frac"] <= 1.0 for e in pr)
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "4 single queries over the full N=120000" in cb["sample"]
    assert out["index_build_sec"] > 0 and out["batched_queries_per_sec"] > 0


@pytest.mark.parametrize("cases,seed,world", [(18, 12008, 2), (8, 12010, 4)])
def test_fuzz_seeds_that_found_unscanned_chunks(cases, seed, world):
    """Regression by seed (tools/fuzz_2rank.py): round 5's first dynamic chunk schedule of the coarse tile scan used sixteen cursor
    groups whatever the grid -- on a shard of 115 / 142 rows x 300 columns (one block: one group) the chunks of the other fifteen
    groups were never handed out, a neighbour went missing and lambda_q was off by 1e-4 (2 ranks, seed 12008, case 17; 4 ranks,
    seed 12010, case 7).  The groups now follow the grid (launch_scan); these seeds stay green."""
    import subprocess

    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_2rank.py"), str(cases), str(seed), str(world)],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    tail = (p.stdout + p.stderr)[-3000:]
    assert "failures in []" in p.stdout, tail
