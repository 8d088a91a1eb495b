"""Diagnostic (make STAMPS=1): phase times inside knn_finish / score_finish from in-kernel 100 MHz stamps."""
import ctypes as C, os, sys, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
import bench, pyarrowspace_amd as asp
from pyarrowspace_amd import _lib
n, d, k, topk = [int(v) for v in (sys.argv[1:5] + ["1000000", "768", "25", "15"][len(sys.argv) - 1:])]
X = bench.make_data(n, d, 42, torch.device("cuda", 0)) if n <= 1_000_000 else None
Q = bench.make_queries(X, 64, 43)
gp = {"eps": bench.calibrate_eps(X, k), "k": k, "topk": topk, "p": 2.0, "sigma": None}
a, g = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, d, d)
L = _lib.load()
L.as_debug_stamps.argtypes = [C.c_void_p]
acc = []
knn_tot = []
for i in range(48):
    a.search(Q[i], g, 0.62)
    st = np.zeros(32, dtype=np.uint64)
    assert L.as_debug_stamps(st.ctypes.data_as(C.c_void_p)) == 0
    if i >= 8: acc.append(st.astype(np.float64) * 0.01)   # us
# rows inside eps = the scan's neighbour candidates (counted after the timed loop: a 3 GB torch temporary between searches
# leaves the finish kernel cold caches and every phase 30 % slower)
for i in range(8, 48):
    qd = torch.from_numpy(Q[i]).to(X.device, torch.float32)
    knn_tot.append(int((((X - qd) ** 2).sum(1) <= gp["eps"] ** 2).sum().item()))
s = np.mean(acc, axis=0)
names_k = ["select_candidates", "exact eval + barrier", "per-candidate keys + barrier", "rank, records", "index order", "lambda"]
names_s = ["select_candidates", "exact eval + scores", "rank + hit records", "a-posteriori check + publish"]
print("knn_finish phases (us):", {nm: round(s[i + 1] - s[i], 2) for i, nm in enumerate(names_k)}, "total", round(s[6] - s[0], 2))
print("score_finish phases (us):", {nm: round(s[17 + i] - s[16 + i], 2) for i, nm in enumerate(names_s)}, "total", round(s[20] - s[16], 2))
print("knn_finish end -> score_finish start (gmin, pick_thr, filter + boundaries):", round(s[16] - s[6], 2))
if s[22] > 0:   # fused tail (fused_finish_kernel): one launch, both phases
    print("fused tail (us): query staging", round(s[0] - s[22], 2), "| knn phase", round(s[6] - s[0], 2), "| barrier behind the gather of the waves' reports", round(s[24] - s[6], 2),
          "| keys", round(s[25] - s[16], 2), "| select", round(s[17] - s[25], 2),
          "| exact eval + scores", round(s[18] - s[17], 2), "| rank", round(s[19] - s[18], 2), "| check + publish", round(s[20] - s[19], 2),
          "| whole kernel", round(s[20] - s[22], 2), "| candidates", np.mean([a_[26] / 0.01 for a_ in acc]))
    print("neighbour candidates of the 40 queries (the scan's eps prefilter):", sorted(int(v) for v in knn_tot))
