import os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from conftest import clustered, calibrate_eps
from oracle import oracle_c
import pyarrowspace_amd as asp
asp.set_debug(True)
n, d = 3000, 64
X = clustered(n, d, nclust=6, seed=4)
gp = {"eps": calibrate_eps(X, 8), "k": 8, "topk": 5, "p": 2.0, "sigma": None}
Xb = X.copy(); Xb[12, 5] = np.inf
ref = oracle_c.OracleIndex(Xb, gp)
for env in ("", "1"):
    if env: os.environ["ARROWSPACE_NO_BAND_PASS"] = "1"
    a, g = asp.ArrowSpaceBuilder.build(gp, Xb)
    ip, ix, v = g.to_csr()
    bad = []
    for i in range(n):
        mine = set(ix[ip[i]:ip[i+1]].tolist()) - {i}
        want = set(ref.indices[ref.indptr[i]:ref.indptr[i+1]].tolist())
        if mine != want: bad.append(i)
    print("band" if not env else "noband", "bad rows", len(bad), bad[:10], g.build_stats(), flush=True)
    if bad:
        i = bad[0]
        print(" row", i, "mine", sorted(set(ix[ip[i]:ip[i+1]].tolist()) - {i})[:12], "want", sorted(ref.indices[ref.indptr[i]:ref.indptr[i+1]].tolist())[:12], "knn want", ref.knn_idx[i])
