#!/bin/bash
set -o pipefail
python - <<'PY' 2>&1 | grep -v amdgpu.ids
import os, sys, time, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import gpu_clustered
import pyarrowspace_amd as asp
for n in (200000, 1000000):
    X = gpu_clustered(n, 768, 11, scale=100.0)
    for gp in ({"eps": 10.0, "k": 25, "topk": 15, "p": 2.0, "sigma": None, "metric": "cosine", "kernel": "rational"},):
        a, g = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, 768, 768)
        st = g.build_stats()
        print(n, "cosine eps=10: build", round(st["total_s"], 3), "s  K2", round(st["mfma_flops"] / st["knn_mfma_s"] / 1e12, 1), "TF/s", "fallback", st["fallback_rows"], "band", st["band_rows"], flush=True)
        q = np.ascontiguousarray(X[5].double().cpu().numpy() * 1.01)
        t0 = time.perf_counter()
        for _ in range(50): a.search(q, g, 0.62)
        print("   search ms", (time.perf_counter() - t0) / 50 * 1e3)
        del a, g
    del X
PY
