"""Probe: which operand each single-query scan read (fp32 / int8 / int8-high) and which search counters moved, on the data of
tests/test_gpu_parity.py::test_int8_image_scan_returns_what_the_fp32_scan_returns; then the same alternating with
ARROWSPACE_SCAN_FP32=1 as that test does.  ARROWSPACE_DEBUG=1 shows why a coarse scan was given up."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import calibrate_eps, clustered
import pyarrowspace_amd as asp
n, d, k, topk = 8000, 200, 12, 9
X = clustered(n, d, nclust=10, seed=31)
gp = {"eps": calibrate_eps(X, k, "l2"), "k": k, "topk": topk, "p": 2.0, "sigma": None}
print("eps", gp["eps"])
aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
rng = np.random.default_rng(3)
Q = [np.ascontiguousarray(X[rng.integers(0, n)] + 0.03 * rng.standard_normal(d) / np.sqrt(d)) for _ in range(24)]
Q.append(np.ascontiguousarray(X[77])); Q.append(np.ascontiguousarray(3.5 * X[1234]))
spike = np.ascontiguousarray(X[5] * 1e-3); spike[0] = 1.0; Q.append(spike)
for tau in (0.62, 1.0):
    for i, q in enumerate(Q):
        c0 = aspace.search_counters()
        try:
            aspace.search(q, gl, tau)
        except asp.PanicException:
            pass
        c1 = aspace.search_counters()
        diff = {k2: c1[k2] - c0[k2] for k2 in c1 if c1[k2] != c0[k2]}
        print(tau, i, aspace.last_scan_operand, diff)
print("---- alternating with ARROWSPACE_SCAN_FP32=1 as the test does")
for tau in (0.62, 1.0, 0.0):
    for i, q in enumerate(Q):
        os.environ.pop("ARROWSPACE_SCAN_FP32", None)
        c0 = aspace.search_counters()
        try:
            got = aspace.search(q, gl, tau)
        except asp.PanicException:
            got = None
        op = aspace.last_scan_operand
        c1 = aspace.search_counters()
        diff = {k2: c1[k2] - c0[k2] for k2 in c1 if c1[k2] != c0[k2]}
        os.environ["ARROWSPACE_SCAN_FP32"] = "1"
        try:
            want = aspace.search(q, gl, tau)
        except asp.PanicException:
            want = None
        os.environ.pop("ARROWSPACE_SCAN_FP32", None)
        if op == "fp32" or got != want or len(diff) > 1:
            print(tau, i, op, diff, "equal" if got == want else "DIFFERENT")
