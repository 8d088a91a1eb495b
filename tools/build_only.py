#!/usr/bin/env python3
"""One index build at N x 768 (for profiling the build kernels).  usage: build_only.py [n] [reps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pyarrowspace_amd as asp  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
X = bench.make_data(n, 768, 42, torch.device("cuda", 0))
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
for _ in range(reps):
    a, g = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, 768, 768)
    st = g.build_stats()
    print("n=%d mfma %.3fs %.1f TF/s refine %.3fs graph %.3fs fallback_rows %d" % (
        n, st["knn_mfma_s"], st["mfma_flops"] / st["knn_mfma_s"] / 1e12, st["refine_s"], st["graph_s"], st["fallback_rows"]))
