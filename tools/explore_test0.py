"""Exploration (not a test): how close do doc-plausible variants of the feature-space lambda (TAUMODE.md, GRAPH_VARIABLES.md)
come to the reference's tau < 1 fixtures tests/test_0.py:39-61?  Reads tests/golden/test0_toy.json.  Result recorded in
DESIGN.md section 3: none of ~6 700 variants reproduces more than one of the three orders."""
import json, itertools, numpy as np
import os
t = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'test0_toy.json')))
X = np.array(t['items']); q = X[2]*1.05
exp = {float(k): v for k, v in t['expected_order'].items()}
cosq = (X@q)/np.sqrt((X*X).sum(1)*(q@q))

def feat_lap(M, eps, k, sigma, p, norm=True, self_in_k=False, sym='union', kern='rational', lap='comb'):
    # M: nodes x dims
    Y = M/np.linalg.norm(M, axis=1, keepdims=True) if norm else M
    n = len(Y)
    nr = np.linalg.norm(Y, axis=1)
    C = (Y@Y.T)/np.outer(nr, nr)
    Dm = 1-np.maximum(0, C)
    W = np.zeros((n, n))
    for i in range(n):
        cand = [(Dm[i, j], j) for j in range(n) if (j != i or self_in_k) and Dm[i, j] <= eps]
        cand.sort()
        for d, j in cand[:k]:
            if j == i: continue
            w = 1/(1+(d/sigma)**p) if kern == 'rational' else np.exp(-0.5*(d/sigma)**p)
            W[i, j] = w
    if sym == 'union': W = np.maximum(W, W.T)
    elif sym == 'mean': W = 0.5*(W+W.T)
    elif sym == 'sum': W = W+W.T
    deg = W.sum(1)
    if lap == 'comb': L = np.diag(deg)-W
    else:
        s = np.where(deg > 0, 1/np.sqrt(np.where(deg > 0, deg, 1)), 0)
        L = np.diag((deg > 0)*1.0) - W*np.outer(s, s)
    return L

def lam(x, L, tau, ordered=True):
    E = (x@L@x)/(x@x)
    Wm = -L.copy(); np.fill_diagonal(Wm, 0); Wm = np.maximum(Wm, 0)
    diff = (x[:, None]-x[None, :])**2
    e = Wm*diff
    if not ordered: e = np.triu(e)
    T = e.sum()
    G = ((e/T)**2).sum() if T > 0 else 0
    G = min(1, max(0, G))
    return tau*E/(E+tau) + (1-tau)*G, E, G

def order(lams, lq, tau):
    s = tau*cosq + (1-tau)/(1+np.abs(lq-lams))
    return list(np.lexsort((np.arange(len(s)), -s))[:3])

def tausel(v, mode):
    if mode == 'median': m = np.median(v)
    elif mode == 'mean': m = np.mean(v)
    return max(m, 1e-9)

res = []
for norm, selfk, sym, kern, lap, ordered, tm, eps, sigma in itertools.product(
        [True, False], [False, True], ['union', 'mean', 'sum'], ['rational', 'gaussian'], ['comb', 'norm'], [True, False],
        ['median', 'mean', 'gmedianE', 'fixed'], [0.05], [0.05]):
    L = feat_lap(X.T, eps, 5, sigma, 2.0, norm, selfk, sym, kern, lap)
    if tm in ('median', 'mean'):
        lams = np.array([lam(x, L, tausel(x, tm), ordered)[0] for x in X])
        lq = lam(q, L, tausel(q, tm), ordered)[0]
    elif tm == 'gmedianE':
        Es = np.array([lam(x, L, 1.0, ordered)[1] for x in X]); t0 = max(np.median(Es), 1e-9)
        lams = np.array([lam(x, L, t0, ordered)[0] for x in X]); lq = lam(q, L, t0, ordered)[0]
    else:
        lams = np.array([lam(x, L, 0.5, ordered)[0] for x in X]); lq = lam(q, L, 0.5, ordered)[0]
    ok = [order(lams, lq, tau) == exp[tau] for tau in (1.0, 0.9, 0.6, 0.55)]
    res.append((sum(ok), norm, selfk, sym, kern, lap, ordered, tm, ok, [order(lams, lq, tau) for tau in (0.9, 0.6, 0.55)], lams.round(5), round(lq, 5)))
res.sort(key=lambda r: -r[0])
for r in res[:25]: print(r)

print("=== round 2: per-vector tau variants")
res=[]
def tausel2(v, mode):
    if mode=='med': m=np.median(v)
    elif mode=='medabs': m=np.median(np.abs(v))
    elif mode=='mean': m=np.mean(v)
    elif mode=='medsq': m=np.median(v*v)
    elif mode=='meansq': m=np.mean(v*v)
    elif mode=='norm': m=np.linalg.norm(v)
    elif mode=='max': m=np.max(v)
    return max(m,1e-9)
for norm, selfk, sym, kern, lap, ordered, tm, k, sig, itemnorm in itertools.product(
        [True, False], [False, True], ['union','mean','sum'], ['rational'], ['comb','norm'], [True, False],
        ['med','medabs','mean','medsq','meansq','norm','max'], [3,4,5,6,24], [0.05, 0.025], [False, True]):
    L = feat_lap(X.T, 0.05, k, sig, 2.0, norm, selfk, sym, kern, lap)
    Xi = X/np.linalg.norm(X,axis=1,keepdims=True) if itemnorm else X
    lams = np.array([lam(x, L, tausel2(x, tm), ordered)[0] for x in Xi])
    lq = lam(q, L, tausel2(q, tm), ordered)[0]
    ok = [order(lams, lq, tau) == exp[tau] for tau in (1.0, 0.9, 0.6, 0.55)]
    res.append((sum(ok), norm, selfk, sym, lap, ordered, tm, k, sig, itemnorm, [order(lams, lq, tau) for tau in (0.9, 0.6, 0.55)], lams.round(4), round(lq,4)))
res.sort(key=lambda r:-r[0])
from collections import Counter
print(Counter(r[0] for r in res))
for r in res[:12]: print(r)
