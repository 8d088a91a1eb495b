#!/usr/bin/env python3
"""A/B of the knn_mfma_kernel variants in ONE process, interleaved rounds (cdna guide rule 24).
usage: python tools/knn_variants.py [n] [rounds] [variants...]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pyarrowspace_amd as asp  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
variants = [int(v) for v in sys.argv[3:]] or list(range(8))
dev = torch.device("cuda", 0)
X = bench.make_data(n, 768, 42, dev)
eps = bench.calibrate_eps(X, 25)
gp = {"eps": eps, "k": 25, "topk": 15, "p": 2.0, "sigma": None}
res = {v: [] for v in variants}
lam0 = None
for r in range(rounds):
    for v in variants:
        os.environ["ARROWSPACE_KNN_VARIANT"] = str(v)
        a, g = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, 768, 768)
        st = g.build_stats()
        res[v].append(st["mfma_flops"] / st["knn_mfma_s"] / 1e12)
        lam = a.lambdas()
        if lam0 is None:
            lam0 = lam
        assert np.array_equal(lam, lam0), "variant %d changed the result" % v
        del a, g
for v in variants:
    print("variant %d: TF/s median %.1f  min %.1f  max %.1f  (refine+fallback rows ok)" % (v, np.median(res[v]), min(res[v]), max(res[v])))
