"""Randomised checks at sizes the small-case fuzzer does not reach: python tools/fuzz_big.py [cases] [seed] [build|search].
  build:  15 000 ... 60 000 items x 8 ... 128 features (hundreds of column tiles: the symmetric pass's sampled thresholds,
          segments and redo paths), the whole index against the oracle's all-pairs build (lambdas, degrees, graph
          pattern), a few searches.
  search: 100 000 ... 400 000 items (several rounds of row chunks per wave of the scan, thousands of score groups); the
          index is built on the GPU only, the searches (single, batched, tau in {1, .62, .3, 0}, near / item / random /
          far queries) against the CPU scorer over the GPU's lambdas and degrees (oracle_c.OracleSearchOnly)."""
import os, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "16")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pyarrowspace_amd as asp
from conftest import assert_hits_match, clustered
from oracle import oracle_c


def make(rng, n, d):
    kind = str(rng.choice(["clustered", "clustered", "gauss", "scaled"]))
    if kind == "clustered":
        X = clustered(n, d, nclust=int(rng.choice([4, 32, 256, 2000])), noise=float(rng.uniform(0.1, 0.6)), seed=int(rng.integers(1 << 30)),
                      normalise=bool(rng.integers(0, 2)))
    else:
        X = rng.standard_normal((n, d))
        if kind == "scaled":
            X *= float(10 ** rng.uniform(-2, 2))
    if rng.random() < 0.3:
        rows = rng.choice(n, int(rng.integers(2, 400)), replace=False)
        X[rows] = X[rows[0]]                                       # a group of exact duplicates
    if rng.random() < 0.5:
        X = X.astype(np.float32).astype(np.float64)
    metric = str(rng.choice(["l2", "cosine"]))
    m = 300
    S = X[rng.choice(n, m, replace=False)]
    if metric == "l2":
        D = np.sqrt(np.maximum(((S[:, None, :] - S[None, :, :]) ** 2).sum(-1), 0))
    else:
        nn = np.linalg.norm(S, axis=1); nn[nn == 0] = 1
        D = 1 - np.maximum(0, (S @ S.T) / np.outer(nn, nn))
    dv = D[np.triu_indices(m, 1)]
    eps = float(np.quantile(dv, float(rng.choice([0.0005, 0.002, 0.01, 0.05, 0.3])))) + 1e-12
    k = int(rng.choice([1, 4, 10, 25, 40, 56, 100]))
    gp = {"eps": eps, "k": k, "topk": int(rng.choice([1, 5, 15, 64, 200])), "p": float(rng.choice([1.0, 2.0])),
          "sigma": None if rng.random() < 0.5 else eps * float(rng.uniform(0.3, 2.0)), "metric": metric,
          "kernel": str(rng.choice(["gaussian", "rational"]))}
    return X, gp, kind


def queries(rng, X):
    n, d = X.shape
    out = []
    for qk in ("near", "item", "random", "far", "near", "item"):
        r = int(rng.integers(n))
        q = {"near": X[r] * 1.01 + 0.01 * rng.standard_normal(d) * (np.abs(X[r]).mean() + 1e-9), "item": X[r].copy(),
             "random": rng.standard_normal(d) * (np.abs(X).mean() + 1e-9), "far": X[r] * 50.0}[qk]
        out.append((np.ascontiguousarray(q), float(rng.choice([1.0, 0.62, 0.3, 0.0]))))
    return out


def check_searches(rng, aspace, gl, ref, X, cfg):
    qs = queries(rng, X)
    for q, tau in qs:
        try:
            want, lq = ref.search(q, tau)
        except oracle_c.ZeroLambda:
            try:
                aspace.search(q, gl, tau)
                raise AssertionError("GPU answered where the oracle panics: %s" % cfg)
            except asp.PanicException:
                continue
        got = aspace.search(q, gl, tau)
        assert_hits_match(got, want, ref.scores(q, tau, lq), rtol=1e-6, atol=1e-9)
    tau = qs[0][1]
    Q = np.stack([q for q, _ in qs] * 7)                            # 42 queries: two passes
    singles = []
    for q in Q:
        try:
            singles.append(aspace.search(np.ascontiguousarray(q), gl, tau))
        except asp.PanicException:
            singles.append(None)
    if all(s is not None for s in singles):
        assert aspace.search_batch(Q, gl, tau) == singles, ("batch != singles", cfg)


def one_case(rng, c, mode):
    if mode == "build":
        n = int(rng.choice([15000, 25000, 40000, 60000]))
        d = int(rng.choice([8, 16, 33, 64, 128]))
        if n * d > 2_600_000:
            d = 32
    else:
        n = int(rng.choice([100_000, 131_073, 200_000, 262_144, 400_000]))
        d = int(rng.choice([8, 16, 32, 48, 64, 100, 200]))
        if n * d > 30_000_000:
            d = 64
    X, gp, kind = make(rng, n, d)
    cfg = dict(case=c, n=n, d=d, kind=kind, gp=gp)
    t0 = time.time()
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    t1 = time.time()
    if mode == "build":
        ref = oracle_c.OracleIndex(X, gp)
        np.testing.assert_allclose(aspace.lambdas(), ref.lambdas, rtol=1e-6, atol=1e-300, err_msg=str(cfg))
        np.testing.assert_allclose(gl.degrees(), ref.deg, rtol=1e-6, atol=1e-300, err_msg=str(cfg))
        indptr, indices, values = gl.to_csr()
        rows = np.repeat(np.arange(n), np.diff(indptr))
        assert np.array_equal(indices[indices != rows], ref.indices), cfg
    else:
        ref = oracle_c.OracleSearchOnly(X, gp, gl.degrees(), aspace.lambdas(), gl.tau0)
    check_searches(rng, aspace, gl, ref, X, cfg)
    st = gl.build_stats()
    print("  case %d ok: n=%d d=%d %s k=%d topk=%d eps=%.3g  gpu build %.2fs (band rows %d, fallback %d), checks %.1fs" % (
        c, n, d, gp["metric"], gp["k"], gp["topk"], gp["eps"], t1 - t0, st.get("band_rows", 0), st.get("fallback_rows", 0), time.time() - t1), flush=True)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    mode = sys.argv[3] if len(sys.argv) > 3 else "build"
    rng = np.random.default_rng(seed)
    bad = 0
    for c in range(cases):
        sub = np.random.default_rng(rng.integers(1 << 62))
        try:
            one_case(sub, c, mode)
        except BaseException as e:   # noqa: BLE001
            bad += 1
            print("FAIL case %d: %s: %s" % (c, type(e).__name__, str(e)[:900]), flush=True)
            if bad >= 6:
                break
    print("fuzz_big (%s): %d cases, %d failures, seed %d" % (mode, c + 1, bad, seed))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
