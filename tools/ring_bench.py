"""One rank's k-NN work on the ring, emulated on one GPU: rank 0 of G holds N/G rows and sees the other shards one at a
time.  Full ring: G blocks (as_knn_block each).  Symmetric ring: the own block, (G-1)//2 whole pairs and, for even G,
half of the opposite pair (as_knn_block_pair) -- plus folding the slices the other ranks would send (not timed: they
arrive from other GPUs).  usage: ring_bench.py N D G"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench
from pyarrowspace_amd.dist import HipEngine, shard_bounds
n, d, G = [int(v) for v in (sys.argv[1:4] + ["1000000", "768", "4"][len(sys.argv) - 1:])]
X = bench.make_data(n, d, 42, torch.device("cuda", 0))
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
if os.environ.get("RING_BENCH_EPS_ALL"):   # the reference's `eps: 10` under the cosine distance (tests/test_3_beir.py:194-200): every pair inside
    gp = {"eps": 10.0, "k": 25, "topk": 15, "p": 2.0, "sigma": None, "metric": "cosine", "kernel": "rational"}
seed_rows = not os.environ.get("RING_BENCH_NO_ROW_THR")
b = shard_bounds(n, G)
counts = [b[i + 1] - b[i] for i in range(G)]
for mode in ("full", "symmetric"):
    e = HipEngine(gp)
    e.create_space(X[b[0]:b[1]].clone())
    e.ring_begin(G)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e.knn_block(e.own_block(), 0, 0, 0)
    t_own = time.perf_counter() - t0
    U = [None] * G
    if mode == "symmetric":
        U0 = e.knn_thresholds(1.0)
        U = [U0] * G      # stand-in thresholds of the right size and kind (every shard is a sample of the same data)
    t1 = time.perf_counter()
    steps = range(1, G) if mode == "full" else range(1, G // 2 + 1)
    for s in steps:
        src = (0 - s) % G
        h = e.open_block(X[b[src]:b[src + 1]].clone())
        if mode == "full":
            e.knn_block(h, src, 0, b[src])
        else:
            row0, row1, ct0, ct1 = 0, counts[0], -1, -1
            if 2 * s == G:
                tq = (counts[src] + 255) // 256 * 256 // 128
                ct0, ct1 = 0, tq // 2
            e.knn_block_pair(h, row0, row1, ct0, ct1, 0, b[src], U[src][: counts[src]], counts[src], row_thr=U[0] if seed_rows else None)
        e.close_block(h)
    torch.cuda.synchronize()
    t_rest = time.perf_counter() - t1
    print(f"G={G} {mode}: own block {t_own:.3f} s, visiting blocks {t_rest:.3f} s, total {t_own + t_rest:.3f} s", flush=True)
    e.close()
