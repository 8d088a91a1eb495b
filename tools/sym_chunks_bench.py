#!/usr/bin/env python3
"""The symmetric k-NN pass in column chunks (a shard whose transposed buffers do not fit a quarter of the free memory:
8M rows at k = 25 -- BASELINE config 5's shard): build time per chunk count at N x 768.
usage: sym_chunks_bench.py N [chunks ...]     (chunks 0 = the library's own choice; "full" = ARROWSPACE_NO_SYM;
"freeG" = the library's own choice with G GB reported free, ARROWSPACE_SYM_FREE_GB: what a ring rank holding five shards sees)"""
import os
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import pyarrowspace_amd as asp  # noqa: E402
from conftest import gpu_clustered  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
cases = sys.argv[2:] or ["0"]
stop = False


def heartbeat():   # a build is one library call of minutes: keep the log moving
    t0 = time.time()
    while not stop:
        time.sleep(30)
        print("  ... %.0f s" % (time.time() - t0), flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
X = gpu_clustered(n, 768, 42)
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
torch.cuda.empty_cache()      # (the calibration's temporaries: the library sizes its scratch by what the driver reports free)
free, total = torch.cuda.mem_get_info()
print("n=%d eps=%.5f free %.1f GB of %.1f" % (n, gp["eps"], free / 1e9, total / 1e9), flush=True)
for c in cases:
    os.environ.pop("ARROWSPACE_SYM_CHUNKS", None)
    os.environ.pop("ARROWSPACE_NO_SYM", None)
    os.environ.pop("ARROWSPACE_SYM_FREE_GB", None)
    if c.startswith("free"):
        os.environ["ARROWSPACE_SYM_FREE_GB"] = c[4:]
    elif c == "full":
        os.environ["ARROWSPACE_NO_SYM"] = "1"
    elif int(c) > 0:
        os.environ["ARROWSPACE_SYM_CHUNKS"] = c
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a, g = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, 768, 768)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = g.build_stats()
    print("chunks=%s: build %.2f s, k-NN kernels %.2f s at %.1f TF/s issued (%.3e flop), refine %.2f s, band rows %d, fallback rows %d"
          % (c, dt, st["knn_mfma_s"], st["mfma_flops"] / st["knn_mfma_s"] / 1e12, st["mfma_flops"], st["refine_s"], st["band_rows"],
             st["fallback_rows"]), flush=True)
    del a, g
    torch.cuda.empty_cache()
stop = True
