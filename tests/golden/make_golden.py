#!/usr/bin/env python3
"""Regenerates tests/golden/*.json|npz.

Two kinds of fixture (data only; no reference source text is stored):
  * reference-pinned literals, re-typed from /root/reference/README.md:37-48,56-62,69
    and /root/reference/tests/test_0.py:4-18,24,29-61 (inputs + expected outputs);
  * oracle vectors: seeded synthetic inputs and the outputs of this repo's own fp64
    oracle (oracle/oracle_np.py, cross-checked against oracle/arrowspace_oracle.c).
    The reference implementation cannot run here (Rust crate not in the tree, no
    cargo), so these pin the SPEC, not the crate.
Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import calibrate_eps, clustered  # noqa: E402
from oracle import oracle_np  # noqa: E402

TAUS = [1.0, 0.8, 0.62, 0.42, 0.0]  # tests/test_2_CVE_db.py:26-28, test_4_msmarco_tau_sweep.py:18-22


def readme_toy():
    return {
        "source": "README.md:37-48,56-62,69",
        "items": [[0.1, 0.2, 0.3], [0.0, 0.5, 0.1], [0.9, 0.1, 0.0]],
        "graph_params": {"eps": 1.0, "k": 6, "topk": 3, "p": 2.0, "sigma": 1.0},
        "query": [0.05, 0.2, 0.25],
        "tau": 1.0,
        "expected_hits": [[0, 0.989743318610787], [1, 0.7565344158360029], [2, 0.22151940739207396]],
    }


def test0_toy():
    items = [
        [0.82, 0.11, 0.43, 0.28, 0.64, 0.32, 0.55, 0.48, 0.19, 0.73, 0.07, 0.36, 0.58, 0.23, 0.44, 0.31, 0.52, 0.16, 0.61, 0.40, 0.27, 0.49, 0.35, 0.29],
        [0.79, 0.12, 0.45, 0.29, 0.61, 0.33, 0.54, 0.47, 0.21, 0.70, 0.08, 0.37, 0.56, 0.22, 0.46, 0.30, 0.51, 0.18, 0.60, 0.39, 0.26, 0.48, 0.36, 0.30],
        [0.78, 0.13, 0.46, 0.27, 0.62, 0.34, 0.53, 0.46, 0.22, 0.69, 0.09, 0.35, 0.55, 0.24, 0.45, 0.29, 0.50, 0.17, 0.59, 0.38, 0.28, 0.47, 0.34, 0.31],
        [0.81, 0.10, 0.44, 0.26, 0.63, 0.31, 0.56, 0.45, 0.20, 0.71, 0.06, 0.34, 0.57, 0.25, 0.47, 0.33, 0.53, 0.15, 0.62, 0.41, 0.25, 0.50, 0.37, 0.27],
        [0.80, 0.12, 0.42, 0.25, 0.60, 0.35, 0.52, 0.49, 0.23, 0.68, 0.10, 0.38, 0.54, 0.21, 0.43, 0.28, 0.49, 0.19, 0.58, 0.37, 0.29, 0.46, 0.33, 0.32],
    ]
    return {
        "source": "tests/test_0.py:4-18,24,29-61",
        "items": items,
        "graph_params": {"eps": 0.05, "k": 5, "topk": 3, "p": 2.0, "sigma": 0.05},
        "query_of_item": 2, "query_scale": 1.05,
        "expected_order": {"1.0": [2, 1, 4], "0.9": [1, 2, 0], "0.6": [1, 3, 2], "0.55": [1, 3, 2]},
        "derivable": ["1.0"],
        "note": "only the tau=1.0 order follows from the documented scorer (pure cosine); the tau<1 orders depend on "
                "lambda values of the un-vendored crate and are recorded as unpinned constraints",
    }


def synth(name, n, d, k, topk, metric, kernel, nclust, seed, nq=6):
    X = clustered(n, d, nclust=nclust, seed=seed)
    eps = calibrate_eps(X, k, metric)
    gp = {"eps": eps, "k": k, "topk": topk, "p": 2.0, "sigma": None, "metric": metric, "kernel": kernel}
    idx = oracle_np.build(X, gp)
    rng = np.random.default_rng(seed + 100)
    Q = np.stack([X[rng.integers(0, n)] + 0.05 * rng.standard_normal(d) / np.sqrt(d) for _ in range(nq)])
    hits_i = np.full((nq, len(TAUS), topk), -1, dtype=np.int64)
    hits_s = np.zeros((nq, len(TAUS), topk))
    lq = np.zeros(nq)
    for a in range(nq):
        for b, tau in enumerate(TAUS):
            h, lq[a] = oracle_np.search(idx, Q[a], tau)
            hits_i[a, b, : len(h)] = [i for i, _ in h]
            hits_s[a, b, : len(h)] = [s for _, s in h]
    np.savez_compressed(
        os.path.join(HERE, name + ".npz"), seed=seed, n=n, d=d, nclust=nclust, eps=eps, k=k, topk=topk,
        metric=metric, kernel=kernel, taus=np.array(TAUS), Q=Q, lambdas=idx["lambdas"], deg=idx["deg"], tau0=idx["tau0"],
        indptr=idx["indptr"], indices=idx["indices"].astype(np.int32), lap=idx["lap"], hits_idx=hits_i, hits_score=hits_s,
        lambda_q=lq)
    print(name, "n", n, "d", d, "nnz", len(idx["indices"]), "tau0", idx["tau0"])


if __name__ == "__main__":
    json.dump(readme_toy(), open(os.path.join(HERE, "readme_toy.json"), "w"), indent=1)
    json.dump(test0_toy(), open(os.path.join(HERE, "test0_toy.json"), "w"), indent=1)
    synth("synth_64x24_l2", 64, 24, 5, 4, "l2", "gaussian", 4, 1)
    synth("synth_400x96_cos", 400, 96, 8, 6, "cosine", "rational", 8, 2)
    synth("synth_1000x384_l2", 1000, 384, 12, 10, "l2", "gaussian", 16, 3)
