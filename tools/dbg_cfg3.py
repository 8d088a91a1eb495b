import os, sys, ctypes as C, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from conftest import gpu_clustered, brute_keys
import pyarrowspace_amd as asp
from pyarrowspace_amd import _lib
L = _lib.load()
asp.set_debug(True)
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 200000, 768
X = gpu_clustered(n, d, 11)
for metric, eps, k in [("cosine", 10.0, 25)]:
    gpd = {"eps": eps, "k": k, "topk": 15, "p": 2.0, "sigma": None, "metric": metric}
    gp, op = asp._parse_graph_params(gpd)
    sp = C.c_void_p()
    assert L.as_space_create_dev(C.c_void_p(X.data_ptr()), _lib.DTYPE_F32, n, d, d, C.byref(op), C.byref(sp)) == 0
    r0, rows = int(sys.argv[3]) if len(sys.argv) > 3 else 1024, int(sys.argv[2]) if len(sys.argv) > 2 else 512
    idx = torch.full((rows, k), -2, dtype=torch.int32, device="cuda"); key = torch.zeros((rows, k), dtype=torch.float64, device="cuda")
    dist = torch.zeros_like(key); gy = torch.zeros_like(key); cnt = torch.zeros(rows, dtype=torch.int32, device="cuda")
    os.environ["ARROWSPACE_NO_BAND_PASS"] = "1"
    assert L.as_knn_rows(sp, C.byref(gp), r0, r0 + rows, C.c_void_p(idx.data_ptr()), C.c_void_p(key.data_ptr()), C.c_void_p(dist.data_ptr()), C.c_void_p(gy.data_ptr()), C.c_void_p(cnt.data_ptr())) == 0
    torch.cuda.synchronize()
    chk = list(range(0, rows, max(1, rows // 512)))
    keys = brute_keys(X, [r0 + t for t in chk], metric)
    vals, bidx = torch.topk(keys, k, dim=1, largest=False)
    ih, bh = idx.cpu().numpy()[chk], bidx.cpu().numpy()
    badrows = [chk[t] for t in range(len(chk)) if set(ih[t].tolist()) != set(bh[t].tolist())]
    bad = len(badrows)
    print(" bad local rows (first 20):", badrows[:20], "row blocks:", sorted(set(b // 256 for b in badrows))[:40])
    rows_ = rows; rows = len(chk)
    print(metric, eps, k, "rows with a different set:", bad, "of", rows, "cnt min/max", int(cnt.min()), int(cnt.max()), flush=True)
    if bad:
        t = next(t for t in range(rows) if set(ih[t].tolist()) != set(bh[t].tolist()))
        print(" row", r0 + chk[t], "ours", ih[t][:6], key[t][:6].cpu().numpy(), "brute", bh[t][:6], vals[t][:6].cpu().numpy())
    L.as_free_space(sp)
