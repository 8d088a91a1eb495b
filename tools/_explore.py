import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import pyarrowspace_amd as asp
from conftest import clustered, calibrate_eps
from oracle import oracle_c

def compare(X, gp, queries, label, taus=(0.62,)):
    t0 = time.time()
    try:
        a, g = asp.ArrowSpaceBuilder.build(gp, X)
    except BaseException as e:
        print("   ", label, "GPU build raised", type(e).__name__, str(e)[:80]); return
    tb = time.time() - t0
    r = oracle_c.OracleIndex(X, gp)
    lam_ok = np.allclose(a.lambdas(), r.lambdas, rtol=1e-9, atol=1e-300)
    res = []
    for q in queries:
        for tau in taus:
            try: w = r.search(q, tau)[0]
            except BaseException as e: w = type(e).__name__
            t1 = time.time()
            try: gq = a.search(np.ascontiguousarray(q), g, tau)
            except BaseException as e: gq = type(e).__name__
            dt = time.time() - t1
            if isinstance(w, str) or isinstance(gq, str): ok = (isinstance(w, str) and isinstance(gq, str))
            else: ok = [i for i, _ in w] == [i for i, _ in gq] and np.allclose([s for _, s in w], [s for _, s in gq], rtol=1e-9)
            res.append((ok, round(dt * 1e3, 2)))
    print("   ", label, "| build %.2fs lambdas ok=%s fallback_rows=%s | queries (ok, ms): %s" % (tb, lam_ok, g.build_stats().get("fallback_rows"), res), flush=True)

n, d = 6000, 64
X = clustered(n, d, nclust=6, seed=5)
gp = {"eps": calibrate_eps(X, 8), "k": 8, "topk": 6, "p": 2.0, "sigma": None}
print("== mass duplicates")
Xd = X.copy(); Xd[:5000] = X[0]
compare(Xd, gp, [X[0] * 1.001, X[5500] * 1.01], "5000 identical rows", taus=(0.62, 1.0))
Xd2 = X.copy(); Xd2[1000:1300] = X[1000]
compare(Xd2, gp, [X[1000] * 1.001, X[2000] * 1.01], "300 identical rows", taus=(0.62, 1.0))
print("== magnitude")
for sc in (1e2, 1e6, 1e18, 1e20, 1e-6, 1e-18, 1e-22):
    gps = dict(gp, eps=gp["eps"] * sc)
    compare(X * sc, gps, [X[3] * sc * 1.01], "items x %g" % sc)
print("== other p / sigma")
compare(X, dict(gp, p=1.0), [X[3] * 1.01], "p=1")
compare(X, dict(gp, p=0.5, sigma=0.3), [X[3] * 1.01], "p=0.5 sigma=0.3")
compare(X, dict(gp, sigma=50.0), [X[3] * 1.01], "sigma=50")
print("== k / topk extremes")
compare(X, dict(gp, k=1, topk=1), [X[3] * 1.01], "k=1 topk=1")
compare(X, dict(gp, k=56, topk=3, eps=gp["eps"] * 1.3), [X[3] * 1.01], "k=56")
print("== tiny d")
for dd in (1, 2, 3):
    Xs = np.abs(clustered(500, dd, nclust=3, seed=6)) + 0.1
    compare(Xs, {"eps": 0.3, "k": 4, "topk": 3, "p": 2.0, "sigma": None}, [Xs[3] * 1.01], "d=%d" % dd)
print("done")
