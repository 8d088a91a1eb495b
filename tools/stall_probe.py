"""When does the one-off stall of the first staged searches happen, and whose is it?  One rank, real RCCL, library-side
exchange: per-query host latency of the first 600 searches after the build -- every query slower than 3 ms is printed
with its index and the time since the first search.  usage: stall_probe.py [N] (environment decides the variant)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import bench
from pyarrowspace_amd import dist as asdist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29545")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000, 768
X = bench.make_data(n, d, 42, torch.device("cuda", 0))
Q = bench.make_queries(X, 600, 43)
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
index = asdist.ShardedIndex.build(gp, X.clone(), dist, force_collectives=True)
if os.environ.get("STALL_PROBE_DESTROY_PG"):
    dist.destroy_process_group()       # the library's communicator is its own: does the stall go with torch's group?
if os.environ.get("STALL_PROBE_SLEEP"):
    time.sleep(float(os.environ["STALL_PROBE_SLEEP"]))
if os.environ.get("STALL_PROBE_GC") == "freeze":
    import gc
    gc.collect(); gc.freeze()          # what the build left behind is out of the collector's sight from here on
elif os.environ.get("STALL_PROBE_GC") == "disable":
    import gc
    gc.disable()
lat, at = [], []
t00 = time.perf_counter()
for q in Q:
    t0 = time.perf_counter(); index.search(q, 0.62); t1 = time.perf_counter()
    lat.append((t1 - t0) * 1e3); at.append((t0 - t00) * 1e3)
slow = [(i, "%.1f ms at +%.0f ms" % (lat[i], at[i])) for i in range(len(lat)) if lat[i] > 3.0]
print("variant %s: library exchange %s; median %.3f ms; total of 600 %.1f ms; slow queries: %s" % (
    os.environ.get("STALL_PROBE_TAG", "default"), bool(index._lib_comm), float(np.median(lat)), sum(lat), slow), flush=True)
