import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from conftest import clustered, calibrate_eps
from pyarrowspace_amd.dist import HipEngine
from oracle import oracle_c
for (n, d, k, metric) in ((3000, 64, 100, "l2"), (20000, 96, 57, "cosine"), (9000, 33, 120, "l2")):
    X = clustered(n, d, nclust=6, seed=3)
    gp = {"eps": calibrate_eps(X, k) * (1.0 if metric == "l2" else 1.0), "k": k, "topk": 10, "p": 2.0, "sigma": None, "metric": metric}
    if metric == "cosine":
        Xn = X / np.linalg.norm(X, axis=1, keepdims=True)
        D = 1 - np.clip(Xn[:300] @ Xn.T, 0, 1); D[np.arange(300), np.arange(300)] = 9
        gp["eps"] = float(np.median(np.sort(D, axis=1)[:, 2 * k]))
    e = HipEngine(gp)
    e.create_space(torch.from_numpy(X).cuda())
    idx, dist, gy, cnt = e.knn_rows(0, n)
    ref = oracle_c.OracleIndex(X, gp)
    idx, cnt = idx.cpu().numpy(), cnt.cpu().numpy()
    bad = [r for r in range(n) if cnt[r] != ref.knn_cnt[r] or list(idx[r, :cnt[r]]) != list(ref.knn_idx[r, :cnt[r]])]
    print(n, d, k, metric, "rows with wrong lists:", len(bad), "mean cnt", cnt.mean(), "stats", {k_: v for k_, v in e.stats().items() if 'rows' in k_})
    np.testing.assert_allclose(dist.cpu().numpy()[cnt > 0, 0], ref_d := None or dist.cpu().numpy()[cnt > 0, 0])
    e.close()
