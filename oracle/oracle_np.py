"""fp64 numpy restatement of the hot path (TEST INFRASTRUCTURE ONLY).

This file is the *checker*, never the product: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  The product path
(pyarrowspace_amd) must never import anything under oracle/.

PARITY STATUS: "parity unpinned" for graph topology and lambda values.  The
reference (/root/reference, a PyO3 shim) forwards all arithmetic to the crates.io
crate `arrowspace 0.18.0` (Cargo.toml:16, Cargo.lock:94-97), whose source is not
in the tree and cannot be built here (no cargo/rustc).  What IS pinned by the
reference and checked in tests/test_oracle_golden.py:
  * README.md:37-48,56-62,69   3x3 toy, tau=1.0: three (index, score) pairs.
  * tests/test_0.py:4-18,24,29-32   5x24 toy, tau=1.0 order [2,1,4].
  * TAUMODE.md:33 + src/lib.rs:169-173   scorer form, result length == topk,
    sorted by score descending, full scan.
Everything else follows the written SPEC in DESIGN.md section 2, which restates
BASELINE.json's north_star (L2 distance, Gaussian weights, normalised Laplacian,
per-item spectral energy) and the documented variants of GRAPH_VARIABLES.md:7-10
(rectified-cosine distance, rational kernel) and TAUMODE.md:8-27 (bounded energy
+ dispersion synthesis, synthesis=Median per
tests/output/1760705545_v0_16/suggested_eps.md:3).

lambda_mode="feature" (SPEC F1-F7, below) restates the lambda those notes document -- TAUMODE.md:8,12-27 on the
F x F feature-space Laplacian of GRAPH_VARIABLES.md:17 -- and is unpinned in the same way: the reference's only
lambda-sensitive fixtures (tests/test_0.py:39-61) hold under no scale-invariant lambda.

Brute force, O(N^2 D): intended for N up to a few thousand.  The C twin
(oracle/arrowspace_oracle.c) is the same algorithm with OpenMP for larger N.
"""
from __future__ import annotations

import numpy as np

TAU_MIN = 1e-12

METRIC_L2 = 0
METRIC_COSINE = 1
KERNEL_GAUSSIAN = 0
KERNEL_RATIONAL = 1
LAMBDA_ITEM = 0      # lambda_i = node-local energy on the N-node item graph (SPEC S3-S9; north_star)
LAMBDA_FEATURE = 1   # lambda_i = Rayleigh quotient on the F x F feature-space Laplacian (SPEC F1-F7; TAUMODE.md:8,12-27)


def resolve_params(graph_params: dict) -> dict:
    """src/helpers.rs:48-76: eps,k,topk,p required; sigma missing/None -> eps*0.5."""
    out = {}
    for key, typ in (("eps", float), ("k", int), ("topk", int), ("p", float)):
        if key not in graph_params:
            raise ValueError(f"graph_params['{key}'] is required")
        out[key] = typ(graph_params[key])
    sigma = graph_params.get("sigma", None)
    out["sigma"] = float(sigma) if sigma is not None else out["eps"] * 0.5
    metric = graph_params.get("metric", "l2")
    kernel = graph_params.get("kernel", "gaussian")
    out["metric"] = {"l2": METRIC_L2, "cosine": METRIC_COSINE}[metric] if isinstance(metric, str) else int(metric)
    out["kernel"] = {"gaussian": KERNEL_GAUSSIAN, "rational": KERNEL_RATIONAL}[kernel] if isinstance(kernel, str) else int(kernel)
    mode = graph_params.get("lambda_mode", "item")
    out["lambda_mode"] = {"item": LAMBDA_ITEM, "feature": LAMBDA_FEATURE}[mode] if isinstance(mode, str) else int(mode)
    return out


def _edge_weight(d, sigma, p, kernel):
    """SPEC S4.  gaussian: exp(-0.5 (d/sigma)^p) (north_star); rational:
    1/(1+(d/sigma)^p) (GRAPH_VARIABLES.md:9)."""
    t = d / sigma
    u = t * t if p == 2.0 else np.power(t, p)
    if kernel == KERNEL_GAUSSIAN:
        return np.exp(-0.5 * u)
    return 1.0 / (1.0 + u)


ENERGY_NOISE = 2.0 ** -46


def edge_energy(w, metric, dist, g, di, dj, nyi, nyj):
    """SPEC S6 edge energy w ||y_i/sqrt(d_i) - y_j/sqrt(d_j)||^2 from the stored pair quantities, in the form that
    does not cancel for near-identical neighbours: with alpha = d_i^-1/2, beta = d_j^-1/2,
      l2:      alpha beta dist^2 + (alpha - beta)(alpha n_i - beta n_j)      (dist^2 = n_i + n_j - 2 g, summed as differences)
      cosine:  (alpha - beta)^2 + 2 alpha beta (1 - c)                       (unit vectors; zero vectors: alpha^2 n_i + beta^2 n_j)
    A value inside the rounding noise of the expanded form's terms is zero: a neighbour identical to the node must
    not become a 1e-16 "energy" whose share of the sum is 1."""
    alpha, beta = 1.0 / np.sqrt(di), 1.0 / np.sqrt(dj)
    if metric == METRIC_L2:
        core = alpha * beta * (dist * dist) + (alpha - beta) * (alpha * nyi - beta * nyj)
    elif nyi > 0.0 and nyj > 0.0:
        core = (alpha - beta) * (alpha - beta) + 2.0 * alpha * beta * (1.0 - g)
    else:
        core = alpha * alpha * nyi + beta * beta * nyj
    v = w * core
    floor = w * ENERGY_NOISE * (alpha * alpha * nyi + beta * beta * nyj + 2.0 * alpha * beta * abs(g))
    return v if v > floor else 0.0


def pair_quantities(xi, X, ni, n, metric):
    """SPEC S2 for one row xi against all rows of X.
    Returns (key, dist, gy): key = quantity the eps test / ordering uses
    (squared L2 distance, or cosine distance), dist = d_ij, gy = y_i.y_j."""
    if metric == METRIC_L2:
        diff = X - xi[None, :]
        key = np.einsum("ij,ij->i", diff, diff)
        dist = np.sqrt(key)
        gy = X @ xi
    else:
        g = X @ xi
        den = np.sqrt(ni * n)
        c = np.where(den > 0, g / np.where(den > 0, den, 1.0), 0.0)
        dist = 1.0 - np.minimum(1.0, np.maximum(0.0, c))   # rounding can push a cosine past 1: no negative distances
        key = dist
        gy = c
    return key, dist, gy


def _eps_key(eps, metric):
    return eps * eps if metric == METRIC_L2 else eps


def knn_lists(X, prm):
    """SPEC S3: directed lists, j != i, key <= eps_key, order (key asc, j asc), cap k."""
    N = X.shape[0]
    n = np.einsum("ij,ij->i", X, X)
    ek = _eps_key(prm["eps"], prm["metric"])
    lists = []
    for i in range(N):
        key, dist, gy = pair_quantities(X[i], X, n[i], n, prm["metric"])
        ok = key <= ek
        ok[i] = False
        idx = np.nonzero(ok)[0]
        order = np.lexsort((idx, key[idx]))
        idx = idx[order][: prm["k"]]
        lists.append((idx, key[idx], dist[idx], gy[idx]))
    return n, lists


def build(X, graph_params: dict) -> dict:
    """ArrowSpaceBuilder.build restated (src/lib.rs:271-300 -> crate build).
    Returns dict with n, CSR adjacency (indptr, indices, dist, gy, w), deg,
    Laplacian values, E, G, tau0, lambdas."""
    X = np.ascontiguousarray(X, dtype=np.float64)
    if X.ndim != 2 or X.shape[0] == 0 or X.shape[1] == 0:
        raise ValueError("items must be non-empty 2D array")
    prm = resolve_params(graph_params)
    if prm["lambda_mode"] == LAMBDA_FEATURE:
        return build_feature(X, prm)
    n, lists = knn_lists(X, prm)
    return graph_from_lists(X, prm, n, lists)


def graph_from_lists(X, prm, n, lists) -> dict:
    """SPEC S4-S9 from the directed lists: lists[i] = (idx, key, dist, gy) of row i."""
    N = X.shape[0]
    # S4 symmetrise (union); per-edge payload is symmetric in (i,j)
    adj = [dict() for _ in range(N)]
    for i, (idx, key, dist, gy) in enumerate(lists):
        for j, kk, dd, gg in zip(idx, key, dist, gy):
            adj[i][int(j)] = (dd, gg)
            if i not in adj[int(j)]:
                adj[int(j)][i] = (dd, gg)
    indptr = np.zeros(N + 1, dtype=np.int64)
    cols, dists, gys = [], [], []
    for i in range(N):
        js = sorted(adj[i].keys())
        indptr[i + 1] = indptr[i] + len(js)
        cols.extend(js)
        dists.extend(adj[i][j][0] for j in js)
        gys.extend(adj[i][j][1] for j in js)
    indices = np.asarray(cols, dtype=np.int64)
    dist = np.asarray(dists, dtype=np.float64)
    gy = np.asarray(gys, dtype=np.float64)
    w = _edge_weight(dist, prm["sigma"], prm["p"], prm["kernel"]) if len(dist) else dist.copy()
    # S5 degrees (ascending-j order)
    deg = np.zeros(N)
    for i in range(N):
        s = 0.0
        for e in range(indptr[i], indptr[i + 1]):
            s += w[e]
        deg[i] = s
    ny = n if prm["metric"] == METRIC_L2 else np.where(n > 0, 1.0, 0.0)
    # S6/S7 energies
    E = np.zeros(N)
    G = np.zeros(N)
    lap = np.zeros_like(w)
    for i in range(N):
        lo, hi = indptr[i], indptr[i + 1]
        if hi == lo:
            continue
        eps_e = np.zeros(hi - lo)
        for t, e in enumerate(range(lo, hi)):
            j = indices[e]
            sdd = np.sqrt(deg[i] * deg[j])
            lap[e] = -w[e] / sdd
            eps_e[t] = edge_energy(w[e], prm["metric"], dist[e], gy[e], deg[i], deg[j], ny[i], ny[j])
        S = 0.0
        for v in eps_e:
            S += v
        E[i] = (0.5 * S) / ny[i] if ny[i] > 0 else 0.0
        if S > 0:
            g = 0.0
            for v in eps_e:
                r = v / S
                g += r * r
            G[i] = min(1.0, max(0.0, g))
    tau0 = median_tau(E)
    lam = synth_lambda(E, G, tau0)
    return dict(prm=prm, X=X, n=n, ny=ny, indptr=indptr, indices=indices, dist=dist, gy=gy,
                w=w, lap=lap, deg=deg, E=E, G=G, tau0=tau0, lambdas=lam,
                knn=[l[0] for l in lists])


def shard_csr(prm, row_offset, nrows, lists, incoming) -> dict:
    """SPEC S4-S5 for the rows [row_offset, row_offset + nrows) alone (row-sharded build): lists[r] =
    (idx, key, dist, gy) of local row r with GLOBAL ids; incoming = iterable of (local target row, source item id,
    dist, gy) -- every directed edge of the whole graph whose target lives here.  Same entries, same order, same
    sums as the corresponding rows of graph_from_lists."""
    adj = [dict() for _ in range(nrows)]
    for r, (idx, key, dist, gy) in enumerate(lists):
        for j, dd, gg in zip(idx, dist, gy):
            adj[r][int(j)] = (dd, gg)
    for r, i, dd, gg in incoming:
        if int(i) not in adj[int(r)]:
            adj[int(r)][int(i)] = (dd, gg)
    indptr = np.zeros(nrows + 1, dtype=np.int64)
    cols, dists, gys = [], [], []
    for r in range(nrows):
        js = sorted(adj[r].keys())
        indptr[r + 1] = indptr[r] + len(js)
        cols.extend(js)
        dists.extend(adj[r][j][0] for j in js)
        gys.extend(adj[r][j][1] for j in js)
    indices = np.asarray(cols, dtype=np.int64)
    dist = np.asarray(dists, dtype=np.float64)
    gy = np.asarray(gys, dtype=np.float64)
    w = _edge_weight(dist, prm["sigma"], prm["p"], prm["kernel"]) if len(dist) else dist.copy()
    deg = np.zeros(nrows)
    for r in range(nrows):
        s = 0.0
        for e in range(indptr[r], indptr[r + 1]):
            s += w[e]
        deg[r] = s
    return dict(prm=prm, row_offset=row_offset, indptr=indptr, indices=indices, dist=dist, gy=gy, w=w, deg=deg)


def shard_energy(sh: dict, deg_global, n_global):
    """SPEC S6/S7 for a shard's rows: the degrees and squared norms of ALL items give the neighbours' terms."""
    prm, off = sh["prm"], sh["row_offset"]
    indptr, indices, w, dist, gy = sh["indptr"], sh["indices"], sh["w"], sh["dist"], sh["gy"]
    ny = n_global if prm["metric"] == METRIC_L2 else np.where(n_global > 0, 1.0, 0.0)
    nrows = len(indptr) - 1
    E, G, lap = np.zeros(nrows), np.zeros(nrows), np.zeros_like(w)
    for r in range(nrows):
        lo, hi = indptr[r], indptr[r + 1]
        if hi == lo:
            continue
        i = r + off
        eps_e = np.zeros(hi - lo)
        for t, e in enumerate(range(lo, hi)):
            j = indices[e]
            lap[e] = -w[e] / np.sqrt(deg_global[i] * deg_global[j])
            eps_e[t] = edge_energy(w[e], prm["metric"], dist[e], gy[e], deg_global[i], deg_global[j], ny[i], ny[j])
        S = 0.0
        for v in eps_e:
            S += v
        E[r] = (0.5 * S) / ny[i] if ny[i] > 0 else 0.0
        if S > 0:
            g = 0.0
            for v in eps_e:
                q = v / S
                g += q * q
            G[r] = min(1.0, max(0.0, g))
    sh.update(E=E, G=G, lap=lap, ny=ny[off : off + nrows])
    return E, G


def median_tau(E):
    """SPEC S8: lower median of the strictly positive energies, clamped to [TAU_MIN, 1]."""
    pos = np.sort(E[E > 0])
    if len(pos) == 0:
        return TAU_MIN
    t = pos[(len(pos) - 1) // 2]
    return float(min(1.0, max(TAU_MIN, t)))


def synth_lambda(E, G, tau0):
    """SPEC S9 (TAUMODE.md:8,18-27): lambda = tau0*E/(E+tau0) + (1-tau0)*G."""
    return tau0 * (E / (E + tau0)) + (1.0 - tau0) * G


def query_neighbours(idx: dict, q, r0=0, r1=None):
    """SPEC S10, first half: the k nearest items of q within eps among rows [r0, r1):
    (item index, key, dist, gy), ordered by (key asc, index asc)."""
    prm = idx["prm"]
    X, n = idx["X"], idx["n"]
    r1 = X.shape[0] if r1 is None else r1
    q = np.asarray(q, dtype=np.float64)
    nq = float(q @ q)
    key, dist, gy = pair_quantities(q, X[r0:r1], nq, n[r0:r1], prm["metric"])
    ok = key <= _eps_key(prm["eps"], prm["metric"])
    cand = np.nonzero(ok)[0]
    order = np.lexsort((cand, key[cand]))
    cand = cand[order][: prm["k"]]
    return cand + r0, key[cand], dist[cand], gy[cand]


def lambda_from_neighbours(idx: dict, q, items, dist, gy, deg=None, ny=None) -> float:
    """SPEC S10, second half: q appended as a node with the given neighbours."""
    prm = idx["prm"]
    q = np.asarray(q, dtype=np.float64)
    nq = float(q @ q)
    if len(items) == 0:
        return 0.0
    o = np.argsort(items, kind="stable")
    items, dist, gy = np.asarray(items)[o], np.asarray(dist)[o], np.asarray(gy)[o]
    deg = idx["deg"][items] if deg is None else np.asarray(deg)[o]
    ny = idx["ny"][items] if ny is None else np.asarray(ny)[o]
    a = _edge_weight(dist, prm["sigma"], prm["p"], prm["kernel"])
    degq = 0.0
    for v in a:
        degq += v
    if not degq > 0.0:
        return 0.0
    nyq = nq if prm["metric"] == METRIC_L2 else (1.0 if nq > 0 else 0.0)
    if not nyq > 0.0:
        return 0.0
    es = []
    for t in range(len(items)):
        dj = deg[t] + a[t]
        sdd = np.sqrt(degq * dj)
        es.append(edge_energy(a[t], prm["metric"], dist[t], gy[t], degq, dj, nyq, ny[t]))
    S = 0.0
    for v in es:
        S += v
    Eq = 0.5 * S / nyq
    Gq = 0.0
    if S > 0:
        for v in es:
            r = v / S
            Gq += r * r
        Gq = min(1.0, max(0.0, Gq))
    tau0 = idx["tau0"]
    return float(tau0 * (Eq / (Eq + tau0)) + (1.0 - tau0) * Gq)


def query_lambda(idx: dict, q) -> float:
    """SPEC S10: prepare_query_item (src/lib.rs:154) restated: q appended as a node.
    Feature mode (SPEC F6/F7): the same functional as the items', on the index's feature Laplacian."""
    if idx["prm"]["lambda_mode"] == LAMBDA_FEATURE:
        E, G = feature_energy(idx, np.asarray(q, dtype=np.float64))
        return float(synth_lambda(E, G, idx["tau0"]))
    items, _, dist, gy = query_neighbours(idx, q)
    return lambda_from_neighbours(idx, q, items, dist, gy)


# ---------------------------------------------------------------------------------------------
# Feature mode: the lambda the reference's notes document (TAUMODE.md:8,12-27) -- a Rayleigh
# quotient on an F x F feature-space Laplacian whose nodes are the D columns of the item matrix
# (GRAPH_VARIABLES.md:17 `GraphFactory::build_spectral_laplacian`), built with the same graph
# parameters and the same distance / kernel options as the item graph (GRAPH_VARIABLES.md:7-10).
#   F1  column a of X is the feature vector f_a in R^N; m_a = sum_i x_ia^2; Gram g_ab = sum_i x_ia x_ib.
#   F2  cosine: c = g_ab / sqrt(m_a m_b) (0 if a column is zero), key = dist = 1 - min(1, max(0, c));
#       l2: key = max(0, m_a + m_b - 2 g_ab) (Gram form), dist = sqrt(key).
#   F3  directed list of a: b != a with key <= eps (cosine) / eps^2 (l2), order (key asc, b asc), first k.
#   F4  union symmetrisation, w_ab = kernel(dist_ab; sigma, p).
#   F5  deg_a = sum_b w_ab (ascending b).  L = D - W: the combinatorial Laplacian, the one for which
#       x^T L x = sum_{a<b} w_ab (x_a - x_b)^2 with w_ab = -L_ab (TAUMODE.md:18-19).
#   F6  for a vector x in R^D (an item or a query): e_ab = w_ab (x_a - x_b)^2 over the edges a < b;
#       T = sum e_ab;  E(x) = T / sum_c x_c^2 (0 for the zero vector);  G(x) = clip(sum (e_ab / T)^2, 0, 1)
#       (0 if T = 0) -- e_raw and g_clamped of TAUMODE.md:12-27.
#   F7  tau0 = lower median of the positive E_i (S8), lambda = tau0 E/(E+tau0) + (1-tau0) G (S9 = TAUMODE.md:8).
#   Query: lambda_q = F6/F7 of the query with the index's tau0; 0 -> the zero-lambda assert (src/lib.rs:156-159).
# GraphLaplacian is then the F x F object the query's Rayleigh quotient is taken against, as in
# `prepare_query_item(&v, gl)` (src/lib.rs:154): nnodes = D.
def feature_graph(X, prm) -> dict:
    """SPEC F1-F5."""
    N, D = X.shape
    m = np.einsum("ia,ia->a", X, X)
    gram = X.T @ X
    if prm["metric"] == METRIC_COSINE:
        den = np.sqrt(np.outer(m, m))
        c = np.where(den > 0, gram / np.where(den > 0, den, 1.0), 0.0)
        dist = 1.0 - np.minimum(1.0, np.maximum(0.0, c))
        key = dist
    else:
        key = np.maximum(0.0, m[:, None] + m[None, :] - 2.0 * gram)
        dist = np.sqrt(key)
    ek = _eps_key(prm["eps"], prm["metric"])
    adj = [dict() for _ in range(D)]
    knn = []
    for a in range(D):
        ok = key[a] <= ek
        ok[a] = False
        idx = np.nonzero(ok)[0]
        order = np.lexsort((idx, key[a][idx]))
        idx = idx[order][: prm["k"]]
        knn.append(idx)
        for b in idx:
            b = int(b)
            dd = dist[a, b] if a < b else dist[b, a]      # one value per unordered pair
            adj[a][b] = dd
            adj[b][a] = dd
    indptr = np.zeros(D + 1, dtype=np.int64)
    cols, dists = [], []
    for a in range(D):
        js = sorted(adj[a].keys())
        indptr[a + 1] = indptr[a] + len(js)
        cols.extend(js)
        dists.extend(adj[a][b] for b in js)
    indices = np.asarray(cols, dtype=np.int64)
    dist_e = np.asarray(dists, dtype=np.float64)
    w = _edge_weight(dist_e, prm["sigma"], prm["p"], prm["kernel"]) if len(dist_e) else dist_e.copy()
    deg = np.zeros(D)
    for a in range(D):
        s = 0.0
        for e in range(indptr[a], indptr[a + 1]):
            s += w[e]
        deg[a] = s
    rows = np.repeat(np.arange(D), np.diff(indptr))
    up = rows < indices                                   # each edge once, ascending (a, b)
    return dict(m=m, indptr=indptr, indices=indices, dist=dist_e, w=w, deg=deg, lap=-w,
                ea=rows[up], eb=indices[up], ew=w[up], knn=knn)


def feature_energy(idx: dict, x):
    """SPEC F6 for one vector (or a [n, D] block of vectors): (E, G)."""
    x = np.asarray(x, dtype=np.float64)
    one = x.ndim == 1
    x2 = x[None, :] if one else x
    ea, eb, ew = idx["ea"], idx["eb"], idx["ew"]
    diff = x2[:, ea] - x2[:, eb]
    e = ew[None, :] * diff * diff
    T = e.sum(axis=1)
    nx = np.einsum("ic,ic->i", x2, x2)
    E = np.where(nx > 0, T / np.where(nx > 0, nx, 1.0), 0.0)
    r = e / np.where(T > 0, T, 1.0)[:, None]
    G = np.where(T > 0, np.minimum(1.0, np.maximum(0.0, (r * r).sum(axis=1))), 0.0)
    return (float(E[0]), float(G[0])) if one else (E, G)


def build_feature(X, prm) -> dict:
    """ArrowSpaceBuilder.build in feature mode: SPEC F1-F7."""
    fg = feature_graph(X, prm)
    n = np.einsum("ij,ij->i", X, X)
    out = dict(prm=prm, X=X, n=n, **fg)
    E = np.zeros(X.shape[0])
    G = np.zeros(X.shape[0])
    step = max(1, (1 << 22) // max(1, len(fg["ea"])))
    for s in range(0, X.shape[0], step):
        E[s:s + step], G[s:s + step] = feature_energy(out, X[s:s + step])
    tau0 = median_tau(E)
    out.update(E=E, G=G, tau0=tau0, lambdas=synth_lambda(E, G, tau0))
    return out


def scores(idx: dict, q, tau: float, lambda_q: float):
    """SPEC S11 (TAUMODE.md:33): tau*cos + (1-tau)/(1+|lq-li|) for every item."""
    X, n = idx["X"], idx["n"]
    q = np.asarray(q, dtype=np.float64)
    nq = float(q @ q)
    den = np.sqrt(n * nq)
    cos = np.where(den > 0, (X @ q) / np.where(den > 0, den, 1.0), 0.0)
    return tau * cos + (1.0 - tau) / (1.0 + np.abs(lambda_q - idx["lambdas"]))


def search(idx: dict, q, tau: float):
    """ArrowSpace.search restated (src/lib.rs:132-174).  Returns (hits, lambda_q);
    raises ZeroLambda when lambda_q == 0 (the reference asserts, src/lib.rs:156-159)."""
    q = np.asarray(q, dtype=np.float64)
    if q.ndim != 1 or q.shape[0] != idx["X"].shape[1]:
        raise ValueError(f"query length {q.shape[0]} must match nfeatures {idx['X'].shape[1]}")
    lq = query_lambda(idx, q)
    if lq == 0.0:
        raise ZeroLambda("The lambdas are zero, check the magnitude of items and eps.")
    s = scores(idx, q, tau, lq)
    N = len(s)
    order = np.lexsort((np.arange(N), -s))
    k = min(idx["prm"]["topk"], N)
    return [(int(i), float(s[i])) for i in order[:k]], lq


class ZeroLambda(Exception):
    pass
