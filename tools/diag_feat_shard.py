"""Diagnosis: feature-mode build over two shards in one process against the whole-space build, stage by stage."""
import os, sys
os.environ["FUZZ_FEATURE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import torch
from fuzz_parity import gen_case
from oracle import oracle_c
from pyarrowspace_amd.dist import HipEngine

seed, want = int(sys.argv[1]), set(int(v) for v in sys.argv[2].split(","))
rng = np.random.default_rng(seed)
for c in range(max(want) + 1):
    sub = np.random.default_rng(rng.integers(1 << 62))
    X, gp, cfg = gen_case(sub, c)
    n, d = X.shape
    if c not in want:
        continue
    split = int(sub.integers(1, n)) if sub.random() < 0.8 else int(sub.choice([1, 2, n - 2, n - 1]))
    print("case", c, cfg, "split", split)
    ref = oracle_c.OracleIndex(X, gp)
    w = HipEngine(gp); w.create_space(torch.from_numpy(X).cuda())
    gw = w.feat_gram(); w.feat_graph(gw); Ew, Gw = w.feat_energy(); w.feat_lambdas_global(Ew.contiguous(), Gw.contiguous(), n, 0)
    lw = w.lambdas()
    print("  whole vs oracle: max rel", float(np.max(np.abs(lw - ref.lambdas) / np.maximum(np.abs(ref.lambdas), 1e-300))), "x64 kept:", bool(w.L.as_space_has_f64(w.sp)) if hasattr(w.L, "as_space_has_f64") else "?")
    es = []
    for lo, hi in ((0, split), (split, n)):
        e = HipEngine(gp); e.create_space(torch.from_numpy(X[lo:hi].copy()).cuda()); es.append(e)
    gs = [e.feat_gram() for e in es]
    g = gs[0].clone(); g += gs[1]
    print("  gram: max abs diff", float((g - gw).abs().max()), "max |gram|", float(gw.abs().max()))
    Es, Gs = [], []
    for e in es:
        e.feat_graph(g)
        E, G = e.feat_energy(); Es.append(E); Gs.append(G)
    E = torch.cat(Es).contiguous(); G = torch.cat(Gs).contiguous()
    print("  E: max rel diff", float(((E - Ew).abs() / Ew.abs().clamp_min(1e-300)).max()), " G: max abs diff", float((G - Gw).abs().max()))
    # same graph? compare CSR of the feature graphs
    import ctypes as C
    def csr(e):
        L = e.L; rows, nnz = int(L.as_nnodes(e.gr)), int(L.as_graph_nnz(e.gr))
        ip, ix, v = np.zeros(rows + 1, dtype=np.int64), np.zeros(nnz, dtype=np.int64), np.zeros(nnz)
        e._check(L.as_graph_csr(e.gr, ip.ctypes.data_as(C.c_void_p), ix.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))
        return ip, ix, v
    a, b = csr(w), csr(es[0])
    print("  feature graph: same pattern", np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), "nnz", len(a[1]), len(b[1]),
          "max val diff", float(np.max(np.abs(a[2] - b[2]))) if len(a[2]) == len(b[2]) else None)
    for e, (lo, hi) in zip(es, ((0, split), (split, n))):
        e.feat_lambdas_global(E, G, n, lo)
        l = e.lambdas()
        print("  shard [%d,%d): lambdas vs whole max rel" % (lo, hi), float(np.max(np.abs(l - lw[lo:hi]) / np.maximum(np.abs(lw[lo:hi]), 1e-300))),
              "tau0", e.tau0(), w.tau0())
