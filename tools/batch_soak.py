#!/usr/bin/env python3
"""as_search_batch against the single-query search at the headline shape: B distinct queries (near items, in-distribution draws, exact
items) through the paired passes (one scan per 64 queries, two pairs of workspaces alternating), every slot equal to what `search`
returns for that query -- hits, scores, and the zero-lambda panics.    python tools/batch_soak.py [N] [D] [B] [tau ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pyarrowspace_amd as asp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
taus = [float(t) for t in sys.argv[4:]] or [0.62, 1.0, 0.3]
dev = torch.device("cuda:0")
X = bench.make_data(n, d, 42, dev)
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, d, d)
rng = np.random.default_rng(7)
rows = rng.integers(0, n, B)
Q = X[torch.from_numpy(rows).to(dev)].double().cpu().numpy()
kinds = rng.integers(0, 3, B)
Q[kinds == 0] += 0.02 * rng.standard_normal((int((kinds == 0).sum()), d)) / np.sqrt(d)
Q[kinds == 1] += 0.3 * rng.standard_normal((int((kinds == 1).sum()), d)) / np.sqrt(d)
Q /= np.linalg.norm(Q, axis=1, keepdims=True)
bad = 0
for tau in taus:
    singles, ok = [], []
    for i in range(B):
        try:
            singles.append(aspace.search(np.ascontiguousarray(Q[i]), gl, tau))
            ok.append(i)
        except asp.PanicException:
            pass
    Qk = np.ascontiguousarray(Q[ok])
    d0 = aspace.batch_dual_scans
    got = aspace.search_batch(Qk, gl, tau)
    miss = [i for i in range(len(ok)) if got[i] != singles[i]]
    bad += len(miss)
    print("tau=%.2f: %d queries (%d with a zero lambda left out), %d shared scans, %d slots differ from the single search%s"
          % (tau, len(ok), B - len(ok), aspace.batch_dual_scans - d0, len(miss), (": first %s" % miss[:5]) if miss else ""), flush=True)
sys.exit(1 if bad else 0)
