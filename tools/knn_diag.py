#!/usr/bin/env python3
"""Diagnostic timing of knn_mfma_kernel ablations (results are WRONG by construction for
variants >= 8; only the kernel time matters).  usage: knn_diag.py [n] variants..."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, pyarrowspace_amd as asp
n = int(sys.argv[1]); variants = [int(v) for v in sys.argv[2:]]
X = bench.make_data(n, 768, 42, torch.device("cuda", 0))
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
for r in range(2):
    for v in variants:
        os.environ["ARROWSPACE_KNN_VARIANT"] = str(v)
        a, g = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, 768, 768)
        st = g.build_stats()
        print("round %d variant %2d: %.1f TF/s" % (r, v, st["mfma_flops"] / st["knn_mfma_s"] / 1e12), flush=True)
        del a, g
