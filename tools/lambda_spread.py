"""Distribution of the item lambdas of the bench index (how wide is [min, max], and without a few outliers?)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import pyarrowspace_amd as asp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
for metric, kernel in (("l2", "gaussian"), ("cosine", "rational")):
    X = bench.make_data(n, d, 42, torch.device("cuda", 0))
    gp = {"eps": bench.calibrate_eps(X, 25, metric), "k": 25, "topk": 15, "p": 2.0, "sigma": None, "metric": metric, "kernel": kernel}
    a, g = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, d, d)
    lam = np.sort(np.asarray(a.lambdas()))
    qs = [0, 1e-6, 1e-5, 1e-4, 1e-3, 0.01, 0.5, 0.99, 1 - 1e-3, 1 - 1e-4, 1 - 1e-5, 1 - 1e-6, 1]
    print(metric, kernel, "tau0 %.4f" % g.tau0, "quantiles:", " ".join("%g:%.4f" % (q, lam[min(n - 1, int(q * (n - 1)))]) for q in qs), flush=True)
    for drop in (0, 16, 64, 256, 1024):
        print("   spread without the %d lowest and %d highest: %.4f" % (drop, drop, lam[n - 1 - drop] - lam[drop]), flush=True)
    del a, g, X
    torch.cuda.empty_cache()
