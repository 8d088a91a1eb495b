#!/usr/bin/env python3
"""bench.py -- queries/sec + index-build seconds on synthetic N x D fp32 (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W            (default N=1M, D=768)
    python -m torch.distributed.run --nproc-per-node G ... bench.py --gpus G ...

One "step" = one ArrowSpace.search() call (one query, full scan over the N items,
lambda_q + blended scorer + top-k) -- the reference's harnesses issue one query per call
(tests/test_4_msmarco_tau_sweep.py:216).  The index build runs once before the timed
region and is reported next to it (index_build_sec).  Inputs are resident in HBM when
either timed region starts.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def make_data(n, d, seed, device, nclust=1024, noise=0.5):
    """SURVEY section 8(d) recipe on the GPU: clustered Gaussians, rows L2-normalised, fp32."""
    import torch

    g = torch.Generator(device=device)
    g.manual_seed(seed)
    C = torch.randn((nclust, d), generator=g, device=device, dtype=torch.float32)
    g.manual_seed(seed + 1000)
    z = torch.randint(0, nclust, (n,), generator=g, device=device)
    X = torch.empty((n, d), device=device, dtype=torch.float32)
    step = 1 << 17
    for s in range(0, n, step):
        e = min(n, s + step)
        X[s:e] = C[z[s:e]] + noise * torch.randn((e - s, d), generator=g, device=device, dtype=torch.float32)
        X[s:e] /= X[s:e].norm(dim=1, keepdim=True)
    return X


def calibrate_eps(X, k, target=2.0, sample=512, seed=5):
    """eps with mean degree before the k-cap ~ target*k, L2 metric."""
    import torch

    n = X.shape[0]
    g = torch.Generator(device=X.device)
    g.manual_seed(seed)
    rows = torch.randperm(n, generator=g, device=X.device)[:sample]
    A = X[rows].double()
    nn = (X.double() ** 2).sum(1) if n <= (1 << 18) else (X * X).sum(1).double()
    G = A @ X.double().T if n <= (1 << 18) else (X[rows] @ X.T).double()
    D2 = (nn[rows][:, None] + nn[None, :] - 2 * G).clamp_min(0)
    D2[torch.arange(len(rows), device=X.device), rows] = float("inf")
    kth = int(min(target * k, n - 1))
    vals = torch.topk(D2, kth, dim=1, largest=False).values[:, -1]
    return float(vals.median().sqrt().item())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", type=int, default=1_000_000)
    ap.add_argument("--d", type=int, default=768)
    ap.add_argument("--k", type=int, default=25)
    ap.add_argument("--topk", type=int, default=15)
    ap.add_argument("--tau", type=float, default=0.62)
    ap.add_argument("--eps", type=float, default=0.0, help="0 = calibrate to mean degree 2k")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-queries", type=int, default=12)
    ap.add_argument("--cpu-build-n", type=int, default=8192)
    args = ap.parse_args()

    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    os.environ["ARROWSPACE_DEVICE"] = str(local_rank)
    dist = None
    force_dist = os.environ.get("ARROWSPACE_BENCH_FORCE_DIST", "") not in ("", "0")   # 1-GPU rehearsal of the N>1 path
    if world > 1 or force_dist:
        import torch.distributed as dist_mod

        dist = dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group(backend="nccl", device_id=device, rank=rank, world_size=world)

    import pyarrowspace_amd as asp

    n, d = args.n, args.d
    X = make_data(n, d, 42, device)
    # queries: perturbed items (as tests/test_0.py:24 does), so every query has neighbours within eps
    nq_total = max(args.steps + args.warmup, 64)
    gq = torch.Generator(device=device)
    gq.manual_seed(43)
    qrows = torch.randint(0, n, (nq_total,), generator=gq, device=device)
    Qd = X[qrows] + 0.05 * 0.5 * torch.randn((nq_total, d), generator=gq, device=device, dtype=torch.float32) / (d ** 0.5) * (d ** 0.5) / 31.0
    Q = (Qd / Qd.norm(dim=1, keepdim=True)).double().cpu().numpy()
    eps = args.eps if args.eps > 0 else calibrate_eps(X, args.k)
    if dist is not None:
        # every rank derives the same queries and eps from the same seeds; broadcast rank 0's anyway so that
        # a bit-level difference between devices can never desynchronise the ranks' control flow
        qt = torch.from_numpy(Q).to(device)
        et = torch.tensor([eps], dtype=torch.float64, device=device)
        dist.broadcast(qt, 0)
        dist.broadcast(et, 0)
        Q = qt.cpu().numpy()
        eps = float(et.item())
    gp = {"eps": eps, "k": args.k, "topk": args.topk, "p": 2.0, "sigma": None}

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------- index build (timed once, inputs resident in HBM)
    single = world == 1 and not force_dist
    if single:
        barrier()
        t0 = time.perf_counter()
        aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, d, d)
        barrier()
        build_s = time.perf_counter() - t0
        searcher = lambda q: aspace.search(q, gl, args.tau)  # noqa: E731
        bstats = gl.build_stats()
    else:
        from pyarrowspace_amd import dist as asdist

        b = asdist.shard_bounds(n, world)
        shard = X[b[rank]:b[rank + 1]].clone()      # this rank's rows; the rest arrives by RCCL all-gather
        del X
        torch.cuda.empty_cache()
        barrier()
        t0 = time.perf_counter()
        index = asdist.ShardedIndex.build(gp, shard, dist, force_collectives=force_dist)
        barrier()
        build_s = time.perf_counter() - t0
        searcher = lambda q: index.search(q, args.tau)  # noqa: E731
        aspace = gl = None
        bstats = index.build_stats()
    bt = torch.tensor([build_s], device=device, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(bt, op=dist.ReduceOp.MAX)
    build_s = float(bt.item())

    # ---------------- search: W warmup + K timed steps
    for i in range(args.warmup):
        searcher(Q[i % len(Q)])
    scan_us = []
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        searcher(Q[(args.warmup + i) % len(Q)])
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt], device=device, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    # kernel-level timing of the dominant kernel (scan_dots) with HIP events on its own stream,
    # taken over a second pass so the event reads do not perturb the timed region
    asp.enable_search_stats(True)
    for i in range(min(args.steps, 50)):
        searcher(Q[(args.warmup + i) % len(Q)])
        scan_us.append(aspace.last_search_stats()["scan_us"] if single else index.last_scan_us())
    scan_ms = float(np.mean(scan_us)) * 1e-3

    # extension, reported next to the headline, never as it: batched search, 32 query
    # slots per pass over the items (GEMM-shaped scan on fp32 MFMA)
    batched_qps = batch_pass_ms = None
    if single and len(Q) >= 64:
        QB = np.ascontiguousarray(np.concatenate([Q[:64]] * 4))    # 256 queries = 8 passes of 32
        aspace.search_batch(QB, gl, args.tau)
        tb = []
        for _ in range(7):
            barrier()
            t1 = time.perf_counter()
            aspace.search_batch(QB, gl, args.tau)
            barrier()
            tb.append(time.perf_counter() - t1)
        batched_qps = len(QB) / float(np.median(tb))
        batch_pass_ms = float(np.median(tb)) / (len(QB) / 32) * 1e3

    qps = args.steps / dt
    rows_per_gpu = (n + world - 1) // world
    scan_bytes = rows_per_gpu * (d + 1) * 4.0          # N x D fp32 read + N fp32 dots written, per launch
    achieved = scan_bytes / (scan_ms * 1e-3) / 1e9
    query_bytes = n * (d + 2) * 4.0                      # whole-query algorithmic bytes
    mfma_tf = bstats["mfma_flops"] / max(bstats["knn_mfma_s"], 1e-9) / 1e12

    # HBM-side traffic per launch from the committed PMC profile (only for the profiled workload)
    traffic_scan = traffic_mfma = traffic_batch = None
    # rows up to 1024 floats take the LDS-DMA ring scan, wider rows the register-staged one
    scan_kernel = "scan_dma_kernel" if d <= 1024 and os.environ.get("ARROWSPACE_SCAN_VARIANT", "0") in ("", "0") else "scan_dots_f32_kernel"
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        if tj["workload"] == {"n": n, "d": d} and world == 1:
            traffic_scan = tj.get(scan_kernel, {}).get("bytes_per_launch")
            traffic_mfma = tj.get("knn_mfma_kernel", {}).get("bytes_per_launch")
            traffic_batch = tj.get("scan_gemm_kernel", {}).get("bytes_per_launch")
    except (OSError, KeyError, ValueError):
        pass

    out = {
        "metric": "queries/sec (single-query search, B=1) at N=%dx D=%d fp32; index-build sec alongside" % (n, d),
        "value": qps,
        "unit": "queries/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "synthetic clustered N=%d x D=%d fp32, k=%d topk=%d tau=%.2f eps=%.5f (mean degree ~2k), "
                               "L2 metric + Gaussian weights; BASELINE.json headline config" % (n, d, args.k, args.topk, args.tau, eps),
                   "n": n, "d": d, "parallelism": "row-shard x%d" % world},
        "index_build_sec": build_s,
        "batched_queries_per_sec": batched_qps,
        "build_stages_sec": {k: bstats[k] for k in ("ingest_s", "knn_mfma_s", "refine_s", "fallback_s", "graph_s")},
        "build_fallback_rows": bstats["fallback_rows"],
        "roofline": {"kernel": scan_kernel, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic_scan,
                     "avg_launch_ms": scan_ms, "bytes_per_launch": scan_bytes},
        "roofline_query": {"bound": "hbm", "achieved": query_bytes / world / (dt / args.steps) / 1e9, "peak": HBM_PEAK_GBS,
                           "unit": "GB/s", "frac": query_bytes / world / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                           "note": "whole query, host-visible latency, per GPU"},
        "roofline_batch": None if batch_pass_ms is None else {
            "kernel": "scan_gemm_kernel + per-slot selection", "bound": "hbm", "queries_per_pass": 32,
            "achieved": query_bytes / (batch_pass_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": query_bytes / (batch_pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "ms_per_pass": batch_pass_ms,
            "traffic": traffic_batch,
            "note": "whole 32-query pass, host-visible; N*(D+2)*4 bytes per pass; fp32 MFMA work 2*32*N*D flops"},
        "roofline_build": {"kernel": "knn_mfma_kernel", "bound": "mfma", "achieved": mfma_tf, "peak": MFMA_F32_PEAK_TF,
                           "unit": "TFLOP/s", "frac": mfma_tf / MFMA_F32_PEAK_TF, "traffic": traffic_mfma,
                           "flops_issued": bstats["mfma_flops"], "kernel_sec": bstats["knn_mfma_s"]},
    }

    # ---------------- CPU baseline: the oracle (fp64 C/OpenMP restatement) on this box's host cores
    if rank == 0 and single and not args.no_cpu_baseline:
        from oracle import oracle_c

        cores = oracle_c.threads()
        Xh = X.double().cpu().numpy()
        ref = oracle_c.OracleSearchOnly(Xh, gp, gl.degrees(), aspace.lambdas(), gl.tau0)
        nq = args.cpu_queries
        ref.search(Q[0], args.tau)
        t0 = time.perf_counter()
        for i in range(nq):
            ref.search(Q[(args.warmup + i) % len(Q)], args.tau)
        cpu_dt = time.perf_counter() - t0
        nb = min(args.cpu_build_n, n)
        t0 = time.perf_counter()
        oracle_c.OracleIndex(Xh[:nb], gp)
        cpu_build = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": nq / cpu_dt, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": "%d single queries over the full N=%d x D=%d fp64 items (two fp64 scans per query: lambda_q k-NN + "
                      "scorer), OpenMP over items" % (nq, n, d),
            "index_build": {"value": cpu_build, "unit": "s", "n": nb,
                            "sample": "all-pairs fp64 build on the first %d rows only (N^2 work: not extrapolated)" % nb},
        }
    if rank == 0:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
