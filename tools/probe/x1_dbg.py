"""Probe: per query, whether the library-issued one-exchange pass ran (as_query_x1_passes) and whether it asked for a redo --
one rank, real RCCL.  python tools/probe/x1_dbg.py N"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch, ctypes as C
import torch.distributed as dist
import bench
import pyarrowspace_amd as asp
from pyarrowspace_amd import dist as asdist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29547")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
n, d = int(sys.argv[1]), 768
X = bench.make_data(n, d, 42, torch.device("cuda", 0))
Q = bench.make_queries(X, 200, 43)
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
index = asdist.ShardedIndex.build(gp, X.clone(), dist, force_collectives=True)
e = index.engine
hist = []
for q in Q[:100]:
    p0 = e.x1_passes(library=True)
    try:
        index.search(q, 0.62)
    except asp.PanicException:
        pass
    hist.append((e.x1_passes(library=True) - p0, int(e.L.as_query_x1_redo(e.qs))))
print("per query (x1 passes, redo flag):", hist[:70])
index.close()
dist.destroy_process_group()
