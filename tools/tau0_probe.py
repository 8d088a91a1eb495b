#!/usr/bin/env python3
"""Which check sends in-distribution queries at small tau to a second pass: tools/tau0_probe.py [n] [tau ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pyarrowspace_amd as asp  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
taus = [float(t) for t in sys.argv[2:]] or [0.0, 0.05, 0.3, 0.62]
X, C = bench.make_data(n, 768, 42, torch.device("cuda", 0), return_centres=True)
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
a, g = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, 768, 768)
Qin = bench.make_queries_in_distribution(C, 300, 43)
Qp = bench.make_queries(X, 300, 43)
for name, Q in (("perturbed-item", Qp), ("in-distribution", Qin)):
    for tau in taus:
        c0 = a.search_counters()
        t0 = time.perf_counter()
        for q in Q[:200]:
            try:
                a.search(q, g, tau)
            except asp.PanicException:
                pass
        dt = time.perf_counter() - t0
        c1 = a.search_counters()
        print(name, "tau=%.2f: %.0f q/s" % (tau, 200 / dt), {k: c1[k] - c0[k] for k in c0}, flush=True)
