"""Per-query host-visible latency of the fused single-GPU search over a few thousand queries: percentiles and the
outliers with their time stamps (is anything periodic stalling the chain?).  usage: latency_trace.py [n] [queries]"""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import pyarrowspace_amd as asp
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
X = bench.make_data(n, 768, 42, torch.device("cuda", 0))
Q = bench.make_queries(X, 512, 43)
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
torch.cuda.synchronize()
aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, 768, 768)
for q in Q[:50]:
    aspace.search(q, gl, 0.62)
for label in ("gc on", "gc off"):
    if label == "gc off":
        gc.disable()
    lat, stamp = np.zeros(nq), np.zeros(nq)
    t00 = time.perf_counter()
    for i in range(nq):
        t0 = time.perf_counter()
        aspace.search(Q[i % 512], gl, 0.62)
        t1 = time.perf_counter()
        lat[i], stamp[i] = (t1 - t0) * 1e6, (t0 - t00) * 1e3
    p = np.percentile(lat, [50, 90, 99, 99.9, 100])
    out = [(round(stamp[i], 1), int(lat[i])) for i in np.nonzero(lat > 2 * p[0])[0]]
    print("%s: median %.0f us, p90 %.0f, p99 %.0f, p99.9 %.0f, max %.0f; mean %.1f; outliers (ms since start, us): %s" % (
        label, p[0], p[1], p[2], p[3], p[4], lat.mean(), out[:30]))
