import os
import sys

import numpy as np
import pytest

# the checker's OpenMP team: on a 128-core host the default team costs 0.3 s per call on these small problems
# (set before the checker's library is loaded; bench.py's cpu_baseline leg does not come through here)
os.environ.setdefault("OMP_NUM_THREADS", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def clustered(n, d, nclust=16, noise=0.5, seed=0, normalise=True):
    """Synthetic clustered-Gaussian embeddings (recipe of SURVEY section 8d, scaled down)."""
    rng = np.random.default_rng(seed)
    C = rng.standard_normal((nclust, d))
    z = rng.integers(0, nclust, n)
    X = C[z] + noise * rng.standard_normal((n, d))
    if normalise:
        X /= np.linalg.norm(X, axis=1, keepdims=True)
    return X


def calibrate_eps(X, k, metric="l2", target=2.0, sample=256, seed=1):
    """eps such that the mean degree before the k-cap is about target*k."""
    rng = np.random.default_rng(seed)
    n = X.shape[0]
    rows = rng.choice(n, size=min(sample, n), replace=False)
    G = X[rows] @ X.T
    nn = np.einsum("ij,ij->i", X, X)
    if metric == "l2":
        D = np.sqrt(np.maximum(nn[rows][:, None] + nn[None, :] - 2 * G, 0.0))
    else:
        D = 1.0 - np.maximum(0.0, G / np.sqrt(nn[rows][:, None] * nn[None, :]))
    D[np.arange(len(rows)), rows] = np.inf
    q = min(1.0, target * k / max(n - 1, 1))
    return float(np.quantile(D[np.isfinite(D)], q))


def calibrate_feature_eps(X, k, metric="cosine", target=2.0):
    """Feature mode (SPEC F1-F3): eps such that a column has about target*k other columns inside it."""
    m = np.einsum("ia,ia->a", X, X)
    G = X.T @ X
    if metric == "l2":
        D = np.sqrt(np.maximum(m[:, None] + m[None, :] - 2 * G, 0.0))
    else:
        den = np.sqrt(np.outer(m, m))
        D = 1.0 - np.minimum(1.0, np.maximum(0.0, np.where(den > 0, G / np.where(den > 0, den, 1.0), 0.0)))
    d = X.shape[1]
    off = D[~np.eye(d, dtype=bool)]
    q = min(1.0, target * k / max(d - 1, 1))
    return float(np.quantile(off, q))


def gpu_clustered(n, d, seed, nclust=1024, noise=0.5, scale=1.0, device="cuda"):
    """Full-size test data on the GPU (torch RNG: seconds instead of the minute numpy needs for 10^9 normals; the
    bench keeps SURVEY 8(d)'s numpy recipe): clustered Gaussians, rows normalised, times `scale`, fp32."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    C = torch.randn((nclust, d), generator=g, device=device, dtype=torch.float32)
    z = torch.randint(0, nclust, (n,), generator=g, device=device)
    X = torch.empty((n, d), device=device, dtype=torch.float32)
    step = 1 << 17
    for s in range(0, n, step):
        e = min(n, s + step)
        X[s:e] = C[z[s:e]] + noise * torch.randn((e - s, d), generator=g, device=device, dtype=torch.float32)
        X[s:e] *= scale / X[s:e].norm(dim=1, keepdim=True)
    return X


def brute_keys(X, rows, metric):
    """fp64 keys (squared L2 distance, or rectified-cosine distance) of the sampled rows against every item,
    on the GPU, by the norm expansion (good to ~1e-12 relative to the norms): [len(rows), N] with the self
    entries at +inf."""
    import torch
    n = X.shape[0]
    r = torch.as_tensor(rows, device=X.device)
    A = X[r].double()
    na = (A * A).sum(1)
    out = torch.empty((len(rows), n), dtype=torch.float64, device=X.device)
    step = 1 << 16
    for s in range(0, n, step):
        B = X[s:s + step].double()
        nb = (B * B).sum(1)
        G = A @ B.T
        if metric == "l2":
            out[:, s:s + step] = (na[:, None] + nb[None, :] - 2 * G).clamp_min(0)
        else:
            out[:, s:s + step] = 1.0 - (G / (na[:, None] * nb[None, :]).sqrt()).clamp(0, 1)
    out[torch.arange(len(rows), device=X.device), r] = float("inf")
    return out


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle_c
    oracle_c.build_lib()
    return oracle_c


def assert_hits_match(got, want, all_scores=None, rtol=1e-9, tie=1e-12, atol=0.0):
    """Returned (index, score) lists against the oracle's.  Indices must be identical except
    where the oracle's scores tie to rounding (|ds| <= tie*|s|): two implementations that sum
    in different orders cannot agree on the order of mathematically equal scores.  With
    `all_scores` (the oracle's score of every item) a tie across the top-k boundary is accepted
    too: every returned score must be right and no left-out item may beat the last returned one."""
    gi, gs = [i for i, _ in got], np.array([s for _, s in got])
    wi, ws = [i for i, _ in want], np.array([s for _, s in want])
    assert len(gi) == len(wi)
    np.testing.assert_allclose(gs, ws, rtol=rtol, atol=atol)   # atol: scores are sums of O(1) terms and can cancel to ~0
    assert all(gs[t] >= gs[t + 1] for t in range(len(gs) - 1))
    if gi == wi:
        return
    for t, (a, b) in enumerate(zip(gi, wi)):
        if a == b:
            continue
        # position t differs: it must sit inside a run of tied oracle scores
        tied = [u for u in range(len(ws)) if abs(ws[u] - ws[t]) <= tie * max(abs(ws[t]), 1e-300)]
        if all_scores is not None:
            assert abs(all_scores[a] - ws[t]) <= tie * max(abs(ws[t]), 1e-300) * 10, (t, a, b)
        else:
            assert len(tied) > 1 and a in [wi[u] for u in tied], (t, gi, wi)
