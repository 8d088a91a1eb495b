"""Where a staged (sharded) single query spends its host time: one rank, real RCCL collectives, N = 1M x 768.
Prints the mean host microseconds of every step of ShardedIndex.search and the end-to-end latency next to the fused
single-GPU as_search."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import bench
import pyarrowspace_amd as asp
from pyarrowspace_amd import dist as asdist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000, 768
X = bench.make_data(n, d, 42, torch.device("cuda", 0))
Q = bench.make_queries(X, 400, 43)
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
index = asdist.ShardedIndex.build(gp, X.clone(), dist, force_collectives=True)
e = index.engine
lat = []
for q in Q[:50]:
    t0 = time.perf_counter(); index.search(q, 0.62); lat.append((time.perf_counter() - t0) * 1e6)
for q in Q[50:350]:
    t0 = time.perf_counter(); index.search(q, 0.62); lat.append((time.perf_counter() - t0) * 1e6)
print("latency of the first queries after the build (us):", " ".join("%.0f" % v for v in lat[:12]), "... medians per 50:",
      " ".join("%.0f" % np.median(lat[i:i + 50]) for i in range(0, 350, 50)), " max per 50:", " ".join("%.0f" % np.max(lat[i:i + 50]) for i in range(0, 350, 50)))
torch.cuda.synchronize()
modes = []
real = e.set_mode
e.set_mode = lambda m: (modes.append(m), real(m))[1]
for rep in range(2):
    t0 = time.perf_counter()
    for q in Q[50:350]:
        index.search(q, 0.62)
    torch.cuda.synchronize()
    print("ShardedIndex.search end to end: %.1f us per query (passes per query %.2f, modes seen %s)" % (
        (time.perf_counter() - t0) / 300 * 1e6, len(modes) / 300, sorted(set(modes))))
    modes.clear()
e.set_mode = real
names = ["ctx enter", "set_mode", "query_scan", "gather knn", "query_lambda", "query_score", "gather hits", "query_finish", "ctx exit"]
import gc
for label in ("gc on", "gc off"):
    if label == "gc off":
        gc.disable()
    NQ = 3000
    T = np.zeros((NQ, len(names)))
    stamps = np.zeros(NQ)
    t00 = time.perf_counter()
    for i in range(NQ):
        q = Q[50 + i % 300]
        t = [time.perf_counter()]
        ctx = torch.cuda.stream(index.stream); ctx.__enter__(); t.append(time.perf_counter())
        e.set_mode(0); t.append(time.perf_counter())
        e.query_scan(q, *index.scan_rows); t.append(time.perf_counter())
        knn_all = index._gather_fixed(e.knn_local); t.append(time.perf_counter())
        e.query_lambda(knn_all); t.append(time.perf_counter())
        e.query_score(0.62); t.append(time.perf_counter())
        hits_all = index._gather_fixed(e.hits_local); t.append(time.perf_counter())
        e.query_finish(hits_all); t.append(time.perf_counter())
        ctx.__exit__(None, None, None); t.append(time.perf_counter())
        T[i] = np.diff(t) * 1e6
        stamps[i] = (t[0] - t00) * 1e3
    tot = T.sum(1)
    print("%s: per query median %.0f us, p99 %.0f, max %.0f, mean %.1f" % (label, np.median(tot), np.percentile(tot, 99), tot.max(), tot.mean()))
    for j, nm in enumerate(names):
        print("  %-14s mean %7.1f  p99 %7.1f  max %9.1f us (host)" % (nm, T[:, j].mean(), np.percentile(T[:, j], 99), T[:, j].max()))
    big = np.nonzero(tot > 3 * np.median(tot))[0]
    print("  outliers (ms since start, us, slowest step):", [(round(stamps[i], 1), int(tot[i]), names[int(np.argmax(T[i]))]) for i in big[:20]])
index.close()
dist.destroy_process_group()
