"""One rank, real RCCL, the exchanges issued by the library (as_query_search_staged): host-visible latency of a sharded query
with the one-exchange pass (default) and with the two-exchange chain (the workspaces' switch off: as_query_set_x1), next to the fused single-space
as_search on the same items.  python tools/x1_overhead.py [N] [D]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import bench
import pyarrowspace_amd as asp
from pyarrowspace_amd import dist as asdist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29546")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
X = bench.make_data(n, d, 42, torch.device("cuda", 0))
Q = bench.make_queries(X, 400, 43)
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
index = asdist.ShardedIndex.build(gp, X.clone(), dist, force_collectives=True)
assert index._lib_comm
aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, d, d)


def run(fn, label):
    for q in Q[:40]:
        try: fn(q)
        except asp.PanicException: pass
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        for q in Q[40:340]:
            try: fn(q)
            except asp.PanicException: pass
        ts.append((time.perf_counter() - t0) / 300 * 1e6)
    print("%-44s %.1f us per query (runs: %s)" % (label, min(ts), " ".join("%.1f" % t for t in ts)))
    return min(ts)


for tau in (0.62,):
    for rnd in range(2):
        f = run(lambda q: aspace.search(q, gl, tau), "fused single space, tau=%.2f" % tau)
        index.engine.x1_set_enabled(True)
        p0 = index.engine.x1_passes(library=True)
        a = run(lambda q: index.search(q, tau), "one rank RCCL, ONE exchange, tau=%.2f" % tau)
        assert index.engine.x1_passes(library=True) > p0
        index.engine.x1_set_enabled(False)
        b = run(lambda q: index.search(q, tau), "one rank RCCL, two exchanges, tau=%.2f" % tau)
        index.engine.x1_set_enabled(True)
        print("  -> over fused: one exchange %+.1f us, two exchanges %+.1f us" % (a - f, b - f))
same = all(index.search(q, 0.62) == aspace.search(q, gl, 0.62) for q in Q[340:380] if True)
print("hits equal to the single space's:", same)
index.close()
dist.destroy_process_group()
