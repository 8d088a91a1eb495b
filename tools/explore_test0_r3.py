"""Exploration (not a test), round 3: lambda families that break lambda(1.05*x) == lambda(x), tried against the
reference's tau < 1 fixtures (tests/test_0.py:34-61; literals in tests/golden/test0_toy.json).

Families:
  (a) F x F Laplacian derived from the ITEM graph (GRAPH_VARIABLES.md:17,35: unit-normalise the items, build the item
      graph, then the feature Laplacian): Xn^T L_item Xn used as it is, sparsified to the k strongest couplings per row,
      or re-Laplacianised; and the column graph of the smoothed signals L_item Xn;
  (b) asymmetric evaluation: items and query normalised / evaluated differently (unit items against a raw query and
      the reverse), items carrying an item-graph lambda (feature-signal Rayleigh quotients mixed by the item's
      squared components) against a feature-Laplacian lambda of the query;
  (c) tau chosen per vector from the vector's own values (7 kinds), or globally from the energies, combined with
      (a) and (b); Rayleigh quotient or the un-normalised quadratic form;
  (d) centroids: every set partition of the 5 items (52), feature graph from the transposed centroid matrix
      (the recorded crate log clusters before it builds: tests/output/1760705545_v0_16/suggested_eps.md:7-11).

Output: a count of variants by the number of tau < 1 orders met (0..3), the best variants of each family with the
measured gaps, written to profiles/r03_test0_families.md by --write.
"""
import itertools
import json
import os
import sys
from collections import Counter, defaultdict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
t = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'test0_toy.json')))
X = np.array(t['items'])
q = X[2] * 1.05
EXP = {float(k): v for k, v in t['expected_order'].items()}
COSQ = (X @ q) / np.sqrt((X * X).sum(1) * (q @ q))
TAUS = (0.9, 0.6, 0.55)
EPS, SIG, P = 0.05, 0.05, 2.0


def unit(M):
    n = np.linalg.norm(M, axis=1, keepdims=True)
    return M / np.where(n > 0, n, 1)


def knn_graph(M, k, sym='union', self_in_k=False):
    """Rectified-cosine eps/k graph over the rows of M (GRAPH_VARIABLES.md:7-10) -> weight matrix."""
    Y = unit(M)
    C = np.clip(Y @ Y.T, -1, 1)
    Dm = 1 - np.maximum(0, C)
    n = len(Y)
    W = np.zeros((n, n))
    for i in range(n):
        cand = sorted((Dm[i, j], j) for j in range(n) if (j != i or self_in_k) and Dm[i, j] <= EPS)
        for d, j in cand[:k]:
            if j != i:
                W[i, j] = 1 / (1 + (d / SIG) ** P)
    if sym == 'union':
        W = np.maximum(W, W.T)
    elif sym == 'mean':
        W = 0.5 * (W + W.T)
    return W


def laplacian(W, kind='comb'):
    deg = W.sum(1)
    if kind == 'comb':
        return np.diag(deg) - W
    s = np.where(deg > 0, 1 / np.sqrt(np.where(deg > 0, deg, 1)), 0)
    return np.diag((deg > 0) * 1.0) - W * np.outer(s, s)


def relap(Mx, k=None, how='neg'):
    """Turn a dense symmetric F x F coupling matrix into a graph Laplacian: off-diagonal weights from the entries."""
    A = Mx.copy()
    np.fill_diagonal(A, 0)
    W = np.maximum(0, -A) if how == 'neg' else np.abs(A)
    if k is not None:
        keep = np.zeros_like(W, dtype=bool)
        for i in range(len(W)):
            keep[i, np.argsort(-W[i], kind='stable')[:k]] = True
        W = np.where(keep | keep.T, W, 0)
    return np.diag(W.sum(1)) - W


def tau_of(v, kind):
    f = {'med': np.median, 'mean': np.mean, 'medabs': lambda a: np.median(np.abs(a)), 'medsq': lambda a: np.median(a * a),
         'meansq': lambda a: np.mean(a * a), 'norm': np.linalg.norm, 'max': np.max}[kind]
    return max(float(f(v)), 1e-9)


def energy(x, L, rayleigh=True):
    T = x @ L @ x
    E = T / (x @ x) if rayleigh else T
    Wm = np.maximum(0, -(L - np.diag(np.diag(L))))
    e = Wm * (x[:, None] - x[None, :]) ** 2
    S = e.sum()
    G = float(np.clip(((e / S) ** 2).sum(), 0, 1)) if S > 0 else 0.0
    return max(float(E), 0.0), G


def synth(E, G, tau):
    return tau * E / (E + tau) + (1 - tau) * G


def orders(lams, lq):
    out = []
    for tau in TAUS:
        s = tau * COSQ + (1 - tau) / (1 + np.abs(lq - lams))
        out.append([int(i) for i in np.lexsort((np.arange(5), -s))[:3]])
    return out


def gaps(lams, lq):
    Lq = 1 / (1 + np.abs(lq - lams))
    return Lq[1] - Lq[2], Lq[3] - Lq[2]      # needs > 3.37e-3 (tau .9) and > 2.23e-3 / 1.81e-3 (tau .6 / .55)


def partitions(s):
    if len(s) == 1:
        yield [s]
        return
    first, rest = s[0], s[1:]
    for p in partitions(rest):
        for i in range(len(p)):
            yield p[:i] + [[first] + p[i]] + p[i + 1:]
        yield [[first]] + p


RES = []            # (n_ok, family, description, orders, lams, lq, gaps)


def evaluate(family, desc, L, LQ=None):
    """All lambda-evaluation variants on the feature Laplacian L (items) / LQ (query; default the same)."""
    LQ = L if LQ is None else LQ
    for ray, inorm, qnorm in itertools.product((True, False), (False, True), (False, True)):
        Xi = unit(X) if inorm else X
        qq = q / np.linalg.norm(q) if qnorm else q
        EG = [energy(x, L, ray) for x in Xi]
        EGq = energy(qq, LQ, ray)
        Es = np.array([e for e, _ in EG])
        pos = Es[Es > 0]
        glob = {'gmedE': max(float(np.median(pos)), 1e-9) if len(pos) else 1e-9,
                'gmeanE': max(float(np.mean(Es)), 1e-9), 'fix.5': 0.5, 'fix1': 1.0}
        for tk in ('gmedE', 'gmeanE', 'fix.5', 'fix1', 'med', 'mean', 'medabs', 'medsq', 'meansq', 'norm', 'max'):
            if tk in glob:
                if ray and inorm == qnorm:
                    continue                      # scale-invariant on both sides: lambda_q == lambda_2, known to fail
                taus = [glob[tk]] * 5
                tq = glob[tk]
            else:
                taus = [tau_of(x, tk) for x in Xi]
                tq = tau_of(qq, tk)
            lams = np.array([synth(e, g, min(tt, 1.0) if tk in ('norm', 'max') and False else tt) for (e, g), tt in zip(EG, taus)])
            lq = synth(*EGq, tq)
            o = orders(lams, lq)
            ok = sum(o[i] == EXP[tau] for i, tau in enumerate(TAUS))
            RES.append((ok, family, f"{desc} rayleigh={ray} items_unit={inorm} query_unit={qnorm} tau={tk}", o,
                        lams.round(5).tolist(), round(float(lq), 5), gaps(lams, lq)))


Xn = unit(X)
W_item = {(k, sym): knn_graph(X, k, sym) for k in (5, 4, 2) for sym in ('union', 'mean')}

# --- (a) feature Laplacians derived from the item graph
for (k, sym), Wi in W_item.items():
    for lk in ('comb', 'norm'):
        Li = laplacian(Wi, lk)
        for src, nm in ((Xn, 'Xn'), (X, 'X')):
            D = src.T @ Li @ src                                       # F x F, PSD
            evaluate('a:XtLX', f"item k={k} {sym} L={lk} {nm}^T L {nm} as is", D)
            for kk in (None, 5, 4):
                for how in ('neg', 'abs'):
                    evaluate('a:relap', f"item k={k} {sym} L={lk} {nm}^T L {nm} relap k={kk} {how}", relap(D, kk, how))
            S = (Li @ src).T                                           # smoothed feature signals: F x N
            for fk in (5, 4):
                for fl in ('comb', 'norm'):
                    evaluate('a:smooth', f"item k={k} {sym} L={lk} column graph of (L {nm}) k={fk} L={fl}",
                             laplacian(knn_graph(S, fk), fl))

# --- (c) on the raw column graph with unit items first (GRAPH_VARIABLES.md:35)
for src, nm in ((Xn, 'Xn'), (X, 'X')):
    for fk, sym, fl, selfk in itertools.product((5, 4, 6), ('union', 'mean'), ('comb', 'norm'), (False, True)):
        evaluate('c:columns', f"column graph of {nm} k={fk} {sym} L={fl} self_in_k={selfk}",
                 laplacian(knn_graph(src.T, fk, sym, selfk), fl))

# --- (d) centroids of every set partition of the items
for part in partitions(list(range(5))):
    if len(part) < 2:
        continue
    for cn in (False, True):
        base = Xn if cn else X
        Cm = np.array([base[g].mean(0) for g in part])
        for fk, fl in itertools.product((5, 4), ('comb', 'norm')):
            evaluate('d:centroids', f"partition {part} unit_items={cn} column graph of centroids k={fk} L={fl}",
                     laplacian(knn_graph(Cm.T, fk), fl))

# --- (b) items carry an item-graph lambda, the query a feature-Laplacian lambda (and the reverse)
LF = laplacian(knn_graph(Xn.T, 5), 'comb')
for (k, sym), Wi in W_item.items():
    for lk in ('comb', 'norm'):
        Li = laplacian(Wi, lk)
        # Rayleigh quotient of every feature signal (a column over the items) on the item graph
        lf = np.array([energy(Xn[:, f], Li)[0] for f in range(X.shape[1])])
        gf = np.array([energy(Xn[:, f], Li)[1] for f in range(X.shape[1])])
        for mix in ('sq', 'abs'):
            wts = (lambda v: v * v) if mix == 'sq' else np.abs
            item_l = np.array([(wts(x) @ lf) / wts(x).sum() for x in X])
            q_same = (wts(q) @ lf) / wts(q).sum()
            for ray, qnorm in itertools.product((True, False), (False, True)):
                qq = q / np.linalg.norm(q) if qnorm else q
                Eq, Gq = energy(qq, LF, ray)
                for tk in ('med', 'mean', 'medabs', 'norm', 'max', 'fix.5'):
                    tq = 0.5 if tk == 'fix.5' else tau_of(qq, tk)
                    for item_side in ('raw', 'synth'):
                        lams = item_l if item_side == 'raw' else np.array(
                            [synth(l, g, tq) for l, g in zip(item_l, [(wts(x) @ gf) / wts(x).sum() for x in X])])
                        lq = synth(Eq, Gq, tq)
                        o = orders(lams, lq)
                        ok = sum(o[i] == EXP[tau] for i, tau in enumerate(TAUS))
                        RES.append((ok, 'b:asym', f"items: feature-signal lambdas on item graph k={k} {sym} L={lk} mixed by {mix} "
                                    f"({item_side}); query: feature Laplacian rayleigh={ray} unit={qnorm} tau={tk}", o,
                                    lams.round(5).tolist(), round(float(lq), 5), gaps(lams, lq)))

by_fam = defaultdict(list)
for r in RES:
    by_fam[r[1]].append(r)
print(f"{len(RES)} variants;", Counter(r[0] for r in RES))
lines = ["# tests/test_0.py tau < 1 orders: lambda families tried (round 3)", "",
         "Generated by `python tools/explore_test0_r3.py --write`.  Wanted: [1,2,0] at tau = 0.9, [1,3,2] at 0.6 and 0.55",
         "(`/root/reference/tests/test_0.py:34-61`), i.e. L1 - L2 > 3.37e-3 and L3 - L2 > 2.23e-3 with L_i = 1/(1+|lambda_q - lambda_i|).", "",
         f"{len(RES)} variants; by number of the three orders met: {dict(sorted(Counter(r[0] for r in RES).items()))}", "",
         "| family | variants | 0 met | 1 met | 2 met | 3 met | best variant (orders at .9/.6/.55; L1-L2, L3-L2) |", "|---|---|---|---|---|---|---|"]
for fam in sorted(by_fam):
    rs = by_fam[fam]
    c = Counter(r[0] for r in rs)
    best = max(rs, key=lambda r: (r[0], min(r[6][0] - 3.37e-3, r[6][1] - 2.23e-3)))
    lines.append(f"| {fam} | {len(rs)} | {c[0]} | {c[1]} | {c[2]} | {c[3]} | {best[2]}: {best[3]}; {best[6][0]:+.2e}, {best[6][1]:+.2e} |")
    print(fam, len(rs), dict(c))
    for r in sorted(rs, key=lambda r: -r[0])[:4]:
        print("   ", r[0], r[2], r[3], r[4], r[5], "gaps %.2e %.2e" % r[6])
full = [r for r in RES if r[0] == 3]
lines += ["", f"Variants meeting all three orders: {len(full)}"]
for r in full[:40]:
    lines.append(f"* {r[1]}: {r[2]} — lambdas {r[4]}, lambda_q {r[5]}")
lines += ["", "## Reading", "",
          "* Every variant that meets all three orders is in family (d): the feature graph built on the columns of a *centroid*",
          "  matrix, for 5 of the 51 partitions of the items into two or more clusters.  None of those partitions is what a",
          "  distance-based clustering of the five items returns ({1,2} is the closest pair, 0.062; the partitions that fit put",
          "  2 with 3, 0.127, or 0 with 4, 0.140), except {1},{0,3},{2,4} (a k-means fixed point) — which fits only with a",
          "  per-vector tau = mean(x), while the recorded crate log says `synthesis=Median`",
          "  (`tests/output/1760705545_v0_16/suggested_eps.md:3`).  With 18 hits in 58 752 variants (0.03 %) over three ordinal",
          "  constraints, these are chance fits, not an identification: no variant is adopted as a mode.",
          "* What the sweep does establish: lambda must depend on the scale of the vector (un-normalised quadratic form, or a",
          "  tau taken from the vector's own values) — every scale-invariant form has lambda_q = lambda_2 and keeps item 2 first;",
          "  with a per-vector tau item 1 overtakes item 2 at tau < 1 in most partitions (the qualitative behaviour of the",
          "  fixture), but the order behind it follows the crate's seeded clustering (`src/lib.rs:282-283`: dims reduction on,",
          "  seed 42), which is not in the tree.  Families (a) item-graph-derived F x F Laplacians and (b) asymmetric",
          "  item/query evaluation reach at most two of the three orders.",
          "* Status of the fixture: **0/3 in every shipped mode**; the three tests stay `xfail(strict)`",
          "  (`tests/test_oracle_golden.py::test_test0_tau_lt1_orders_feature_mode`)."]
if '--write' in sys.argv:
    open(os.path.join(ROOT, 'profiles', 'r03_test0_families.md'), 'w').write("\n".join(lines) + "\n")
