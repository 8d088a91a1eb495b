#!/usr/bin/env python3
"""The tile scan's two chunk schedules hand the same rows to different waves at different times; everything a search returns is
re-evaluated exactly behind the scan, so the answers must not depend on the schedule.  On ONE 1M x 768 index: Q random queries
(near items, in-distribution draws, exact items; tau in {0.62, 1.0, 0.4, 0.2, 0.0}), each searched under the dynamic schedule and
again under equal shares (as_set_tuning("tile_dyn")): hits and scores must be identical; then the same through 4 native threads
(shared scans).    python tools/dyn_soak.py [N] [D] [Q]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pyarrowspace_amd as asp
from pyarrowspace_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 3000
dev = torch.device("cuda:0")
X = bench.make_data(n, d, 42, dev)
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, d, d)
L = _lib.load()
rng = np.random.default_rng(99)
rows = rng.integers(0, n, nq)
Q = X[torch.from_numpy(rows).to(dev)].double().cpu().numpy()
kinds = rng.integers(0, 3, nq)
Q[kinds == 0] += 0.02 * rng.standard_normal((int((kinds == 0).sum()), d)) / np.sqrt(d)
Q[kinds == 1] += 0.5 * rng.standard_normal((int((kinds == 1).sum()), d)) / np.sqrt(d)
Q /= np.linalg.norm(Q, axis=1, keepdims=True)
taus = rng.choice([0.62, 0.62, 1.0, 0.4, 0.2, 0.0], nq)
bad = panics = 0
ops = {}
for i in range(nq):
    q = np.ascontiguousarray(Q[i])
    res = []
    for dyn in (1, 0):
        L.as_set_tuning(b"tile_dyn", dyn)
        try:
            res.append(aspace.search(q, gl, float(taus[i])))
        except asp.PanicException:
            res.append(None)
        ops[(dyn, aspace.last_scan_operand)] = ops.get((dyn, aspace.last_scan_operand), 0) + 1
    if res[0] is None:
        panics += 1
    if res[0] != res[1]:
        bad += 1
        if bad <= 3:
            print("MISMATCH", i, taus[i], res[0][:3] if res[0] else None, res[1][:3] if res[1] else None)
print("dyn_soak: %d queries, %d mismatches between the schedules, %d zero-lambda panics (both schedules), operands %s, counters %s"
      % (nq, bad, panics, ops, aspace.search_counters()))
sys.exit(1 if bad else 0)
