"""debug: one case of the seeded larger-size fuzz run (tests/test_gpu_fuzz.py::test_seeded_fuzz_at_larger_sizes), verbose"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_big as fb
import pyarrowspace_amd as asp
from oracle import oracle_c
want_case = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rng0 = np.random.default_rng(20260106)
for case in range(6):
    rng = np.random.default_rng(rng0.integers(1 << 62))
    if case != want_case:
        continue
    n = int(rng.choice([15000, 25000, 40000, 60000])); d = int(rng.choice([8, 16, 33, 64, 128]))
    if n * d > 2_600_000: d = 32
    X, gp, kind = fb.make(rng, n, d)
    print("cfg", n, d, kind, gp, flush=True)
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    ref = oracle_c.OracleIndex(X, gp)
    asp.set_debug(True)
    for q, tau in fb.queries(rng, X):
        try:
            want, lq = ref.search(q, tau)
        except oracle_c.ZeroLambda:
            want, lq = None, 0.0
        try:
            got = aspace.search(q, gl, tau)
        except asp.PanicException:
            got = None
        print("tau", tau, "want", None if want is None else want[:3], "lq", lq, "got", None if got is None else got[:3], flush=True)
