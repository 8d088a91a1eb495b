"""Probe: the coarse scan on a 30 000-row index (the data of test_coarse_scan_returns_what_the_two_digit_scan_returns): operand and
counters per query; ARROWSPACE_DEBUG=1 prints the overflow bits of a scan whose candidates did not fit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import calibrate_eps, clustered
import pyarrowspace_amd as asp
n, d, k, topk = 30000, 256, 12, 9
X = clustered(n, d, nclust=150, seed=71)
gp = {"eps": calibrate_eps(X, k, "l2"), "k": k, "topk": topk, "p": 2.0, "sigma": None, "metric": "l2"}
aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
rng = np.random.default_rng(11)
Q = [np.ascontiguousarray(X[rng.integers(0, n)] + 0.03 * rng.standard_normal(d) / np.sqrt(d)) for _ in range(14)]
for q in Q:
    try:
        aspace.search(q, gl, 0.62)
    except asp.PanicException:
        print("zero lambda")
    print(aspace.last_scan_operand, aspace.search_counters())
