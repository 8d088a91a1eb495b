"""Randomised check of the row-sharded index with 2 (or 3, 4 ... 6) ranks on ONE GPU (collectives staged through the
CPU, gloo): python tools/fuzz_2rank.py [cases] [seed] [world].  The configurations of tools/fuzz_parity.py, random cuts
of the items over the ranks (sometimes a handful of rows on one side); lambdas, k-NN lists of every rank and a few
searches (single and batched) against the oracle.  One set of processes runs all the cases."""
import os, socket, sys, time
os.environ.setdefault("OMP_NUM_THREADS", "8")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np


def worker(rank, world, port, cases, seed, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pyarrowspace_amd as asp
    from conftest import assert_hits_match
    from fuzz_parity import gen_case
    from oracle import oracle_c
    from test_gpu_dist import _cpu_staged
    Staged = _cpu_staged()
    rng = np.random.default_rng(seed)
    bad, t0 = [], time.time()
    for c in range(cases):
        sub = np.random.default_rng(rng.integers(1 << 62))
        X, gp, cfg = gen_case(sub, c)
        n, d = X.shape
        if n < 4 or d > 1024:
            continue
        if n < world + 2:
            continue
        if world == 2:
            split = int(sub.integers(1, n)) if sub.random() < 0.8 else int(sub.choice([1, 2, n - 2, n - 1]))
            bounds = [0, split, n]
        else:
            inner = np.sort(sub.choice(np.arange(1, n), size=world - 1, replace=False))
            if sub.random() < 0.25:
                inner[0] = int(sub.integers(1, min(4, int(inner[1]))))     # a rank with a few rows
            bounds = [0] + [int(v) for v in inner] + [n]
        cfg = dict(cfg, split=bounds[1:-1])
        qs = [np.ascontiguousarray(X[int(sub.integers(n))] * 1.01), np.ascontiguousarray(X[int(sub.integers(n))]),
              np.ascontiguousarray(sub.standard_normal(d) * (np.abs(X).mean() + 1e-9))]
        taus = [float(sub.choice([1.0, 0.62, 0.0])) for _ in qs]
        if os.environ.get("FUZZ_VERBOSE"):
            print("rank %d case %d: %s" % (rank, c, cfg), flush=True)
        shard = X[bounds[rank]:bounds[rank + 1]].copy()
        if sub.random() < 0.5 and np.array_equal(X.astype(np.float32).astype(np.float64), X):
            shard = shard.astype(np.float32)      # fp32 device shards (what bench.py hands over): same index
        index = Staged.build(gp, torch.from_numpy(shard).cuda(), dist)
        lam = index.lambdas()
        feature = gp.get("lambda_mode") == "feature"     # (no item k-NN lists in feature mode)
        lists = None if feature else [t.cpu().numpy().copy() for t in index.engine.lists()]
        got = []
        for q, tau in zip(qs, taus):
            try:
                got.append(index.search(q, tau))
            except asp.PanicException:
                got.append("panic")
        try:
            gotb = index.search_batch(np.stack(qs), taus[0])
        except asp.PanicException:
            gotb = "panic"
        reload_ok = True
        if sub.random() < 0.25:    # one file per rank, loaded again: same shard bounds, same answers
            prefix = os.path.join(os.environ.get("TMPDIR", "/tmp"), "as_fuzz_%d_%d" % (port, c))
            index.save(prefix)
            dist.barrier()
            loaded = Staged.load(prefix, gp, dist)
            again = []
            for q, tau in zip(qs, taus):
                try:
                    again.append(loaded.search(q, tau))
                except asp.PanicException:
                    again.append("panic")
            reload_ok = again == got and (loaded.n, loaded.r0, loaded.r1) == (index.n, index.r0, index.r1)
            loaded.close()
            os.remove("%s.rank%dof%d" % (prefix, rank, world))
        index.close()
        # ---- checks (no collective below this line: a failure must not leave the other rank waiting)
        try:
            assert reload_ok, "the loaded index answers differently"
            ref = oracle_c.OracleIndex(X, gp)
            lo, hi = bounds[rank], bounds[rank + 1]
            if lists is not None:
                np.testing.assert_array_equal(lists[3], ref.knn_cnt[lo:hi])
            np.testing.assert_allclose(lam, ref.lambdas, rtol=1e-6, atol=1e-300)
            wants = []
            for q, tau, g in zip(qs, taus, got):
                try:
                    want, lq = ref.search(q, tau)
                except oracle_c.ZeroLambda:
                    assert g == "panic", "answered where the oracle panics"
                    wants.append("panic")
                    continue
                assert g != "panic", "panicked where the oracle answers"
                assert_hits_match(g, want, ref.scores(q, tau, lq), rtol=1e-6, atol=1e-9)
                wants.append(g)
            if "panic" in [wants[i] for i in range(len(qs))]:
                pass   # (a batch with a zero-lambda query panics as a whole)
            elif gotb != "panic":
                for i, tau in enumerate(taus):
                    if tau == taus[0]:
                        assert gotb[i] == got[i], "batch != single"
        except BaseException as e:   # noqa: BLE001
            bad.append(c)
            import traceback
            tb = traceback.extract_tb(e.__traceback__)[-1]
            print("FAIL rank %d case %d: %s at %s:%d (%s): %s\n   %s" % (rank, c, type(e).__name__, os.path.basename(tb.filename), tb.lineno,
                                                                      tb.line, str(e)[:500], cfg), flush=True)
        if rank == 0 and c % 20 == 19:
            print("  ... %d cases, %d failures on rank 0, %.0fs" % (c + 1, len(bad), time.time() - t0), flush=True)
    out[rank] = bad
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    assert 2 <= world <= 6   # a GPU box takes at most 6 processes on its card
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(worker, args=(world, port, cases, seed, out), nprocs=world, join=True)
    bad = sorted(set().union(*[set(out[r]) for r in range(world)]))
    print("fuzz_2rank: %d cases on %d ranks, failures in %s, seed %d" % (cases, world, bad, seed))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
