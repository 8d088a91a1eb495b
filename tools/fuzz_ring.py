"""Randomised check of the ring build on one GPU: python tools/fuzz_ring.py [cases] [seed].
The configurations of tools/fuzz_parity.py (shapes, metric, eps quantile, duplicates, fp32 / fp64 items), the items cut
into 2-5 random blocks (some tiny); every rank's lists through the plain ring (as_knn_block / band / exact rounds) and
through the symmetric ring (as_knn_block_pair, optionally in forced column chunks) against as_knn_rows on one space
holding everything: same counts and ids; distances to the last bits."""
import os, sys, time, traceback
os.environ.setdefault("OMP_NUM_THREADS", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
from fuzz_parity import gen_case
from test_gpu_ring import _ring_lists, _single_lists, _symmetric_ring_lists


def one_case(rng, case):
    X, gp, cfg = gen_case(rng, case)
    gp = {k: v for k, v in gp.items() if k != "lambda_mode"}
    n = X.shape[0]
    if n < 3:
        return
    G = int(rng.integers(2, min(5, n) + 1))
    inner = np.sort(rng.choice(np.arange(1, n), size=G - 1, replace=False))
    cuts = [0] + [int(v) for v in inner] + [n]
    cfg = dict(cfg, cuts=cuts)
    chunk = int(rng.choice([0, 0, 1, 3, 8]))
    if chunk:
        os.environ["ARROWSPACE_PAIR_CHUNK_TILES"] = str(chunk)
    try:
        res, _ = _symmetric_ring_lists(X, gp, cuts)
    finally:
        os.environ.pop("ARROWSPACE_PAIR_CHUNK_TILES", None)
    plain_block = int(rng.integers(0, G))
    for b in range(G):
        lo, hi = cuts[b], cuts[b + 1]
        sidx, sdist, sgy, scnt = _single_lists(X, gp, lo, hi)
        checks = [("symmetric", res[b])]
        if b == plain_block:
            idx, key, dist, gy, cnt, nflag, over = _ring_lists(X, gp, cuts, b)
            checks.append(("plain", (idx, dist, gy, cnt)))
        for name, (idx, dist, gy, cnt) in checks:
            ctx = "%s ring, block %d, chunk %d: %s" % (name, b, chunk, cfg)
            np.testing.assert_array_equal(cnt, scnt, err_msg=ctx)
            np.testing.assert_array_equal(idx, sidx, err_msg=ctx)
            np.testing.assert_allclose(dist, sdist, rtol=1e-13, atol=1e-300, err_msg=ctx)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    only = set(int(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 and sys.argv[3] != "-" else None
    rng = np.random.default_rng(seed)
    bad, t0 = 0, time.time()
    for c in range(cases):
        sub = np.random.default_rng(rng.integers(1 << 62))
        if only is not None and c not in only:
            continue
        try:
            one_case(sub, c)
        except BaseException as e:   # noqa: BLE001
            bad += 1
            print("FAIL case %d: %s: %s" % (c, type(e).__name__, str(e)[:700]), flush=True)
            if only is not None:
                traceback.print_exc()
            if bad >= 8:
                break
        if c % 20 == 19:
            print("  ... %d cases, %d failures, %.0fs" % (c + 1, bad, time.time() - t0), flush=True)
    print("fuzz_ring: %d cases, %d failures, seed %d" % (c + 1, bad, seed))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
