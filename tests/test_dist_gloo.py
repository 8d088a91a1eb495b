"""CPU, world_size 2, gloo: the row-sharded host logic of pyarrowspace_amd.dist
(shard bounds, padded all-gathers of uneven shards, record exchange, flag agreement,
identical results on every rank) with the CPU oracle injected as the per-rank engine.
The product engine (HipEngine) needs a GPU and is covered by tests/test_gpu_dist.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import calibrate_eps, clustered  # noqa: E402


class OracleEngine:
    """Per-rank engine on numpy (test infrastructure): same interface and the same
    fixed-size record layout (include/arrowspace_hip.h as_knn_rec / as_hit_rec) as HipEngine."""

    def __init__(self, graph_params):
        from oracle import oracle_np
        self.o = oracle_np
        self.prm = oracle_np.resolve_params(graph_params)
        self.mode = 0

    def create_space(self, X):
        self.X = X.double().numpy().copy()
        self.n, self.d = self.X.shape
        self.nn = np.einsum("ij,ij->i", self.X, self.X)

    def knn_rows(self, r0, r1):
        import torch
        k = self.prm["k"]
        rows = r1 - r0
        idx = np.full((rows, k), -1, dtype=np.int32)
        dist = np.zeros((rows, k))
        gy = np.zeros((rows, k))
        cnt = np.zeros(rows, dtype=np.int32)
        ek = self.o._eps_key(self.prm["eps"], self.prm["metric"])
        for i in range(r0, r1):
            key, dd, gg = self.o.pair_quantities(self.X[i], self.X, self.nn[i], self.nn, self.prm["metric"])
            ok = key <= ek
            ok[i] = False
            c = np.nonzero(ok)[0]
            c = c[np.lexsort((c, key[c]))][:k]
            idx[i - r0, : len(c)] = c
            dist[i - r0, : len(c)] = dd[c]
            gy[i - r0, : len(c)] = gg[c]
            cnt[i - r0] = len(c)
        return torch.from_numpy(idx), torch.from_numpy(dist), torch.from_numpy(gy), torch.from_numpy(cnt)

    # ---- ring build (no replication): the same steps as HipEngine's, on numpy
    def ring_begin(self, nblocks):
        self.parts, self.off = {}, 0

    def open_block(self, X):
        return X.double().numpy().copy()

    def close_block(self, h):
        pass

    def own_block(self):
        return self.X

    def block_nmax(self, h):
        return float(np.einsum("ij,ij->i", h, h).max()) if len(h) else 0.0

    def knn_block(self, h, b, row_goff, col_goff):
        k, ek = self.prm["k"], self.o._eps_key(self.prm["eps"], self.prm["metric"])
        nh = np.einsum("ij,ij->i", h, h)
        out = []
        for i in range(self.n):
            if len(h) == 0:
                out.append([])
                continue
            key, dd, gg = self.o.pair_quantities(self.X[i], h, self.nn[i], nh, self.prm["metric"])
            ok = key <= ek
            if col_goff <= row_goff + i < col_goff + len(h):
                ok[row_goff + i - col_goff] = False
            c = np.nonzero(ok)[0]
            c = c[np.lexsort((c, key[c]))][:k]
            out.append([(key[j], col_goff + j, dd[j], gg[j]) for j in c])
        self.parts[b] = out
        self.off = row_goff

    def knn_block_band(self, h, b, row_goff, col_goff):
        return 0

    # ---- symmetric ring: a pair of blocks computed once, both sides' slices from the same distances
    def knn_thresholds(self, nmax_all):
        import torch
        return torch.full((self.n,), float("inf"), dtype=torch.float32)

    def _wi(self):
        k = self.prm["k"]
        return k + 2 + ((k + 2) & 1)            # int32 columns (ids, count, bound bits), padded to an even number

    def slice_shape(self, nrows):
        return (nrows, 3 * self.prm["k"] + self._wi() // 2)

    def knn_block_pair(self, h, row0, row1, ct0, ct1, row_goff, col_goff, col_thr, ncols, row_thr=None):
        import torch
        k, ek = self.prm["k"], self.o._eps_key(self.prm["eps"], self.prm["metric"])
        j0, j1 = (0, ncols) if ct0 < 0 else (min(ncols, ct0 * 128), min(ncols, ct1 * 128))
        nh = np.einsum("ij,ij->i", h, h)
        own = [[] for _ in range(self.n)]
        for i in range(row0, row1):
            key, dd, gg = self.o.pair_quantities(self.X[i], h[j0:j1], self.nn[i], nh[j0:j1], self.prm["metric"])
            c = np.nonzero(key <= ek)[0]
            c = c[np.lexsort((c, key[c]))][:k]
            own[i] = [(key[j], col_goff + j0 + j, dd[j], gg[j]) for j in c]
        self.parts[("pair", col_goff, j0)] = own
        F = np.zeros((ncols, 3, k))
        I = np.full((ncols, self._wi()), -1, dtype=np.int32)
        I[:, k] = 0
        I[:, k + 1] = np.array([np.inf], dtype=np.float32).view(np.int32)[0]
        for j in range(j0, j1):
            key, dd, gg = self.o.pair_quantities(h[j], self.X[row0:row1], nh[j], self.nn[row0:row1], self.prm["metric"])
            c = np.nonzero(key <= ek)[0]
            c = c[np.lexsort((c, key[c]))][:k]
            F[j, 0, : len(c)], F[j, 1, : len(c)], F[j, 2, : len(c)] = key[c], dd[c], gg[c]
            I[j, : len(c)] = row_goff + row0 + c
            I[j, k] = len(c)
        self.off = row_goff
        return torch.from_numpy(np.concatenate([F.reshape(ncols, 3 * k), np.ascontiguousarray(I).view(np.float64)], axis=1))

    def fold_slice(self, P, nmax_src):
        k = self.prm["k"]
        P = P.numpy()
        F = P[:, : 3 * k].reshape(P.shape[0], 3, k)
        I = np.ascontiguousarray(P[:, 3 * k :]).view(np.int32)
        self.parts[("recv", len(self.parts))] = [[(F[i, 0, t], int(I[i, t]), F[i, 1, t], F[i, 2, t]) for t in range(I[i, k])]
                                                  for i in range(self.n)]

    def knn_merge(self, nmax):
        import torch
        k = self.prm["k"]
        idx = np.full((self.n, k), -1, dtype=np.int32)
        dist = np.zeros((self.n, k))
        gy = np.zeros((self.n, k))
        cnt = np.zeros(self.n, dtype=np.int32)
        for i in range(self.n):
            allc = sorted(c for b in self.parts for c in self.parts[b][i])[:k]
            for t, (kk, j, dd, gg) in enumerate(allc):
                idx[i, t], dist[i, t], gy[i, t] = j, dd, gg
            cnt[i] = len(allc)
        self._lists = tuple(torch.from_numpy(a) for a in (idx, dist, gy, cnt))
        return 0

    def lists(self):
        return self._lists

    def norms(self):
        return __import__("torch").from_numpy(self.nn.copy())

    def graph_from_knn_global(self, n_global, row_offset, idx, dist, gy, cnt, n64):
        idx, dist, gy, cnt, n64 = idx.numpy(), dist.numpy(), gy.numpy(), cnt.numpy(), n64.numpy()
        lists = [(idx[i, : cnt[i]].astype(np.int64), dist[i, : cnt[i]], dist[i, : cnt[i]], gy[i, : cnt[i]]) for i in range(n_global)]
        Xg = np.zeros((n_global, self.d))
        Xg[row_offset : row_offset + self.n] = self.X          # only this rank's rows are ever read (scans are local)
        self.index = self.o.graph_from_lists(Xg, self.prm, n64, lists)
        self.off, self.nlocal = row_offset, self.n
        self.n = n_global

    # ---- graph stage without replication (oracle_np.shard_csr / shard_energy)
    def graph_shard_csr(self, n_global, row_offset, idx, dist, gy, cnt, in_row, in_col, in_dist, in_gy):
        idx, dist, gy, cnt = idx.numpy(), dist.numpy(), gy.numpy(), cnt.numpy()
        lists = [(idx[i, : cnt[i]].astype(np.int64), dist[i, : cnt[i]], dist[i, : cnt[i]], gy[i, : cnt[i]]) for i in range(self.n)]
        inc = zip(in_row.numpy(), in_col.numpy(), in_dist.numpy(), in_gy.numpy())
        self.shard = self.o.shard_csr(self.prm, row_offset, self.n, lists, inc)
        self.n_global, self.off, self.nlocal = n_global, row_offset, self.n
        return __import__("torch").from_numpy(self.shard["deg"].copy())

    def graph_shard_energy(self, deg_global, n64_global):
        self.deg_g, self.n64_g = deg_global.numpy().copy(), n64_global.numpy().copy()
        E, _ = self.o.shard_energy(self.shard, self.deg_g, self.n64_g)
        return __import__("torch").from_numpy(E.copy())

    def graph_shard_lambdas(self, E_global):
        tau0 = self.o.median_tau(E_global.numpy())
        lam = np.zeros(self.n_global)
        lam[self.off : self.off + self.nlocal] = self.o.synth_lambda(self.shard["E"], self.shard["G"], tau0)
        Xg = np.zeros((self.n_global, self.d))
        Xg[self.off : self.off + self.nlocal] = self.X          # only this rank's rows are ever read (scans are local)
        ny = self.n64_g if self.prm["metric"] == self.o.METRIC_L2 else np.where(self.n64_g > 0, 1.0, 0.0)
        self.index = dict(prm=self.prm, X=Xg, n=self.n64_g, ny=ny, deg=self.deg_g, tau0=tau0, lambdas=lam)
        self.n = self.n_global

    def graph_from_knn(self, idx, dist, gy, cnt):
        self.off = 0
        idx, dist, gy, cnt = idx.numpy(), dist.numpy(), gy.numpy(), cnt.numpy()
        lists = [(idx[i, : cnt[i]].astype(np.int64), None, dist[i, : cnt[i]], gy[i, : cnt[i]]) for i in range(self.n)]
        lists = [(a, d, d, g) for a, _, d, g in lists]
        self.index = self.o.graph_from_lists(self.X, self.prm, self.nn, lists)

    def query_open(self):
        import torch
        self.k, self.topk = self.prm["k"], min(self.prm["topk"], self.n)
        self.knn_local = torch.zeros((self.k, 6), dtype=torch.float64)
        self.hits_local = torch.zeros((self.topk + 1, 2), dtype=torch.float64)

    def set_mode(self, mode):
        self.mode = mode

    def query_scan(self, q, r0, r1):
        self.q = np.asarray(q, dtype=np.float64)
        r0, r1 = r0 + self.off, r1 + self.off                  # local rows of a shard-only space -> item ids
        self.r0, self.r1 = r0, r1
        items, key, dist, gy = self.o.query_neighbours(self.index, self.q, r0, r1)
        rec = np.zeros((self.k, 6))
        rec[:, 0] = np.array([-1], dtype=np.int64).view(np.float64)[0]
        rec[:, 1] = np.inf
        m = len(items)
        rec[:m, 0] = items.astype(np.int64).view(np.float64)
        rec[:m, 1], rec[:m, 2], rec[:m, 3] = key, dist, gy
        rec[:m, 4], rec[:m, 5] = self.index["deg"][items], self.index["ny"][items]
        self.knn_local.copy_(__import__("torch").from_numpy(rec))

    def query_lambda(self, knn_all):
        r = knn_all.numpy()
        ids = r[:, 0].copy().view(np.int64)
        ok = ids >= 0
        ids, key = ids[ok], r[ok, 1]
        o = np.lexsort((ids, key))[: self.k]
        sel = np.nonzero(ok)[0][o]
        self.lq = self.o.lambda_from_neighbours(self.index, self.q, ids[o], r[sel, 2], r[sel, 3], r[sel, 4], r[sel, 5])

    def query_score(self, tau):
        rec = np.zeros((self.topk + 1, 2))
        rec[:, 0] = np.array([-1], dtype=np.int64).view(np.float64)[0]
        rec[:, 1] = -np.inf
        if self.r1 > self.r0:
            s = self.o.scores(self.index, self.q, tau, self.lq)[self.r0 : self.r1]
            o = np.lexsort((np.arange(len(s)), -s))[: self.topk]
            rec[: len(o), 0] = (o + self.r0).astype(np.int64).view(np.float64)
            rec[: len(o), 1] = s[o]
        rec[self.topk, 0] = np.array([-2], dtype=np.int64).view(np.float64)[0]
        rec[self.topk, 1] = 0.0
        self.hits_local.copy_(__import__("torch").from_numpy(rec))

    def query_finish(self, hits_all):
        r = hits_all.numpy()
        ids = r[:, 0].copy().view(np.int64)
        ok = ids >= 0
        ids, sc = ids[ok], r[ok, 1]
        o = np.lexsort((ids, -sc))[: self.topk]
        hits = [(int(ids[t]), float(sc[t])) for t in o]
        return hits, self.lq, self.lq == 0.0, False, False

    # ---- batched staged search: the single-query steps slot by slot, in the [slot][...] record layout
    def batch_open(self):
        import torch
        self.cap = 4
        self.knn_local_b = torch.zeros((self.cap, self.k, 6), dtype=torch.float64)
        self.hits_local_b = torch.zeros((self.cap, self.topk + 1, 2), dtype=torch.float64)
        return self.cap

    def query_scan_batch(self, Q, r0, r1):
        self.bq, self.brange = [np.asarray(q, dtype=np.float64) for q in Q], (r0, r1)
        empty = np.zeros((self.k, 6))
        empty[:, 0] = np.array([-1], dtype=np.int64).view(np.float64)[0]
        empty[:, 1] = np.inf
        for b in range(self.cap):
            if b < len(self.bq):
                self.query_scan(self.bq[b], r0, r1)
                self.knn_local_b[b].copy_(self.knn_local)
            else:
                self.knn_local_b[b].copy_(__import__("torch").from_numpy(empty))

    def query_lambda_batch(self, knn_all, nranks):
        r = knn_all.reshape(nranks, self.cap, self.k, 6)
        self.blq = []
        for b in range(len(self.bq)):
            self.q = self.bq[b]
            self.query_lambda(r[:, b].reshape(nranks * self.k, 6))
            self.blq.append(self.lq)

    def query_score_batch(self, tau):
        for b in range(len(self.bq)):
            self.q, self.lq = self.bq[b], self.blq[b]
            self.r0, self.r1 = self.brange[0] + self.off, self.brange[1] + self.off
            self.query_score(tau)
            self.hits_local_b[b].copy_(self.hits_local)

    def query_finish_batch(self, hits_all, nranks, nb):
        r = hits_all.reshape(nranks, self.cap, self.topk + 1, 2)
        out = []
        for b in range(nb):
            self.lq = self.blq[b]
            hits, lq, zero, _, _ = self.query_finish(r[:, b].reshape(nranks * (self.topk + 1), 2))
            out.append((hits, lq, zero))
        return out

    def lambdas(self):
        lam = self.index["lambdas"]
        return lam[self.off : self.off + self.nlocal] if getattr(self, "nlocal", None) is not None and self.off + self.nlocal <= len(lam) and hasattr(self, "parts") else lam

    def close(self):
        pass


class OracleEngineX1(OracleEngine):
    """The one-exchange pass of a sharded query (as_query_x1_begin / _finish, include/arrowspace_hip.h) restated on the oracle:
    a rank's block = its k-NN records, a header (count, flags) and its scorer candidates as (item id, exact cosine, lambda) --
    the local rows within W = (1 - tau) / (2 tau) of the rank's topk-th largest cosine, the scan's rule -- so that
    ShardedIndex.search's one-exchange branch (one all-gather, the same finish on every rank) runs under gloo."""
    CAP = 64

    def x1_usable(self, tau):
        return 0.4 <= tau <= 1.0 and not getattr(self, "x1_off", False) and getattr(self, "x1_agreed", True)

    def x1_enabled(self):
        """as_query_x1_enabled: this rank's own switch (its environment when the workspace was made)."""
        return getattr(self, "x1_local_switch", True)

    def x1_set_enabled(self, enabled):
        """as_query_set_x1: what the ranks agreed on."""
        self.x1_agreed = bool(enabled)

    def x1_set_coarse(self, allowed):
        """as_query_set_coarse: the retry of a pass whose candidates did not fit scans both digits (here: a scripted failure
        of the 'coarse' first attempt -- x1_coarse_fails -- goes away with it)."""
        self.x1_coarse_allowed = bool(allowed)
        self.x1_coarse_log = getattr(self, "x1_coarse_log", []) + [bool(allowed)]

    def x1_begin(self, q, tau, r0, r1, world):
        import torch
        self.x1_calls = getattr(self, "x1_calls", 0) + 1
        self.query_scan(q, r0, r1)
        blk = np.zeros((self.k * 6 + 2 + self.CAP * 3,))
        blk[: self.k * 6] = self.knn_local.numpy().reshape(-1)
        flags, cand = 0, np.zeros((0,), dtype=np.int64)
        if self.r1 > self.r0:
            cos = self.o.scores(self.index, self.q, 1.0, 0.0)[self.r0 : self.r1]        # tau = 1: the cosine itself
            kth = np.sort(cos)[::-1][min(self.topk, len(cos)) - 1]
            cand = np.nonzero(cos >= kth - (1.0 - tau) / (2.0 * tau) - 1e-12)[0]
            coarse_fail = getattr(self, "x1_coarse_fails", False) and getattr(self, "x1_coarse_allowed", True)
            if len(cand) > self.CAP or getattr(self, "x1_force_redo", False) or coarse_fail:
                flags, cand = 16, cand[:0]                                              # did not fit: every rank reruns on the chain
            c = blk[self.k * 6 + 2 :].reshape(self.CAP, 3)
            c[: len(cand), 0] = (cand + self.r0).astype(np.int64).view(np.float64)
            c[: len(cand), 1] = cos[cand]
            c[: len(cand), 2] = self.index["lambdas"][cand + self.r0]
        blk[self.k * 6], blk[self.k * 6 + 1] = float(len(cand)), float(flags)
        self.x1_send = torch.from_numpy(blk)
        return self.x1_send

    def x1_finish(self, blocks_all, tau, world):
        r = blocks_all.numpy().reshape(world, -1)
        self.query_lambda(__import__("torch").from_numpy(np.ascontiguousarray(r[:, : self.k * 6].reshape(-1, 6))))
        ids, cs, lam, flags = [], [], [], 0
        for w in range(world):
            cnt, fl = int(r[w, self.k * 6]), int(r[w, self.k * 6 + 1])
            flags |= fl
            c = r[w, self.k * 6 + 2 :].reshape(self.CAP, 3)[:cnt]
            ids.append(c[:, 0].copy().view(np.int64)); cs.append(c[:, 1]); lam.append(c[:, 2])
        ids, cs, lam = np.concatenate(ids), np.concatenate(cs), np.concatenate(lam)
        sc = tau * cs + (1.0 - tau) / (1.0 + np.abs(self.lq - lam))
        o = np.lexsort((ids, -sc))[: self.topk]
        hits = [(int(ids[t]), float(sc[t])) for t in o]
        return hits, self.lq, self.lq == 0.0, False, 0, bool(flags & 16)


def _x1_worker(rank, world, port, n, d, cuts, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyarrowspace_amd.dist import ShardedIndex
        X = clustered(n, d, nclust=6, seed=21)
        gp = {"eps": calibrate_eps(X, 6), "k": 6, "topk": 5, "p": 2.0, "sigma": None}
        bounds = sorted([0, n] + list(cuts))
        shard = torch.from_numpy(X[bounds[rank] : bounds[rank + 1]].copy())
        eng = OracleEngineX1(gp)
        index = ShardedIndex.build(gp, shard, dist, engine=eng)
        rng = np.random.default_rng(5)
        res, calls = [], []
        for i in range(4):
            q = X[rng.integers(0, n)] + 0.02 * rng.standard_normal(d) / np.sqrt(d)
            for tau in (1.0, 0.62, 0.4, 0.2):
                eng.x1_force_redo = i == 3 and rank == world - 1     # one rank's candidates "do not fit": every rank takes the chain
                eng.x1_coarse_fails = i == 2 and rank == 0           # rank 0's "coarse scan" overflows: ONE retry on the fine scan serves it
                before = getattr(eng, "x1_calls", 0)
                one = index.search(q, tau)
                calls.append(getattr(eng, "x1_calls", 0) - before)
                eng.x1_off = True
                two = index.search(q, tau)
                eng.x1_off = False
                assert one == two, (tau, one, two)
                res.append((one, index.last_lambda_q))
        out[rank] = (res, calls)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,cuts", [(2, (130,)), (3, (70, 70))])
def test_one_exchange_search_under_gloo(world, cuts):
    """ShardedIndex.search's one-exchange branch (tau in [0.4, 1]): one all-gather of the ranks' blocks, the same finish on
    every rank -- same hits as the two-exchange chain and as one process, an EMPTY shard included; tau = 0.2 keeps the chain;
    a rank whose candidates do not fit sends every rank to the chain."""
    import torch.multiprocessing as mp
    from oracle import oracle_np
    n, d = 300, 24
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_x1_worker, args=(world, _free_port(), n, d, cuts, out), nprocs=world, join=True)
    X = clustered(n, d, nclust=6, seed=21)
    gp = {"eps": calibrate_eps(X, 6), "k": 6, "topk": 5, "p": 2.0, "sigma": None}
    ref = oracle_np.build(X, gp)
    rng = np.random.default_rng(5)
    want = []
    for _ in range(4):
        q = X[rng.integers(0, n)] + 0.02 * rng.standard_normal(d) / np.sqrt(d)
        for tau in (1.0, 0.62, 0.4, 0.2):
            want.append(oracle_np.search(ref, q, tau))
    for rank in range(world):
        res, calls = out[rank]
        # the one-exchange pass ran for tau >= 0.4 only; the query whose first attempt "overflowed" on one rank (i == 2) and the one
        # whose candidates never fit (i == 3) took a second pass on every rank -- the latter then the two-exchange chain
        # (tau = 0.4: a window of 0.75 in cosine -- more candidates than the test engine's block holds: a second pass, then the chain)
        assert calls == [1, 1, 2, 0] * 2 + [2, 2, 2, 0] * 2
        for (hits, lq), (whits, wlq) in zip(res, want):
            assert [i for i, _ in hits] == [i for i, _ in whits]
            np.testing.assert_allclose([s for _, s in hits], [s for _, s in whits], rtol=1e-12)
            assert abs(lq - wlq) <= 1e-12 * abs(wlq)
        assert out[rank][0] == out[0][0]


def _x1_vote_worker(rank, world, port, n, d, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyarrowspace_amd.dist import ShardedIndex
        X = clustered(n, d, nclust=6, seed=21)
        gp = {"eps": calibrate_eps(X, 6), "k": 6, "topk": 5, "p": 2.0, "sigma": None}
        shard = torch.from_numpy(X[rank * n // world : (rank + 1) * n // world].copy())
        eng = OracleEngineX1(gp)
        eng.x1_local_switch = rank != 1        # rank 1's environment has the one-exchange pass switched off
        index = ShardedIndex.build(gp, shard, dist, engine=eng)
        rng = np.random.default_rng(5)
        res = []
        for _ in range(3):
            q = X[rng.integers(0, n)] + 0.02 * rng.standard_normal(d) / np.sqrt(d)
            res.append(index.search(q, 0.62))
        out[rank] = (eng.x1_agreed, getattr(eng, "x1_calls", 0), res)
    finally:
        dist.destroy_process_group()


def test_one_exchange_switch_is_agreed_over_the_ranks():
    """What a rank's environment switches about the query path decides which collectives a search issues (one all-gather of
    blocks, or two of records): ShardedIndex agrees on it once when the query is opened (MIN over the ranks,
    _agree_query_switches) -- a rank with ARROWSPACE_STAGED_X1=0 switches the pass off for all, no rank ever takes the
    one-exchange branch alone (it would wait in an all-gather nobody else enters), and the hits are those of one process."""
    import torch.multiprocessing as mp
    from oracle import oracle_np
    n, d, world = 300, 24, 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_x1_vote_worker, args=(world, _free_port(), n, d, out), nprocs=world, join=True)
    X = clustered(n, d, nclust=6, seed=21)
    gp = {"eps": calibrate_eps(X, 6), "k": 6, "topk": 5, "p": 2.0, "sigma": None}
    ref = oracle_np.build(X, gp)
    rng = np.random.default_rng(5)
    want = [oracle_np.search(ref, X[rng.integers(0, n)] + 0.02 * rng.standard_normal(d) / np.sqrt(d), 0.62)[0] for _ in range(3)]
    for rank in range(world):
        agreed, calls, res = out[rank]
        assert agreed is False and calls == 0
        for hits, whits in zip(res, want):
            assert [i for i, _ in hits] == [i for i, _ in whits]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, d, split, out, mode="ring"):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyarrowspace_amd.dist import ShardedIndex
        X = clustered(n, d, nclust=6, seed=21)
        gp = {"eps": calibrate_eps(X, 6), "k": 6, "topk": 5, "p": 2.0, "sigma": None}
        bounds = [0, split, n] if world == 2 else ([0, n] if world == 1 else sorted([0, n] + list(split)))
        shard = torch.from_numpy(X[bounds[rank] : bounds[rank + 1]].copy())
        replicate = mode == "replicated"
        index = ShardedIndex.build(gp, shard, dist, engine=OracleEngine(gp), replicate=replicate, gather_lists=mode == "ring_lists")
        assert (index.r0, index.r1) == (bounds[rank], bounds[rank + 1]) and index.n == n and index.replicated == replicate
        assert hasattr(index.engine, "shard") == (mode == "ring")       # the sharded graph stage ran / did not run
        assert getattr(index, "ring_symmetric", False) == (mode != "replicated" and world > 1)   # every block pair computed once
        rng = np.random.default_rng(5)
        res = []
        for _ in range(4):
            q = X[rng.integers(0, n)] + 0.02 * rng.standard_normal(d) / np.sqrt(d)
            for tau in (1.0, 0.62, 0.0):
                res.append((index.search(q, tau), index.last_lambda_q))
        # batched search: 4 slots per pass in the test engine, 9 queries = 3 passes, one partly filled
        rng = np.random.default_rng(6)
        Qb = np.stack([X[rng.integers(0, n)] + 0.02 * rng.standard_normal(d) / np.sqrt(d) for _ in range(9)])
        assert index.search_batch(Qb, 0.62) == [index.search(q, 0.62) for q in Qb]
        out[rank] = (index.lambdas().copy(), res)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["ring", "ring_lists", "replicated"])
@pytest.mark.parametrize("split", [97, 150, 0])
def test_two_rank_sharded_index_matches_single_process(split, mode):
    """ring: every rank keeps only its rows, the shards visit over send/recv (uneven and EMPTY shards included), the
    edges reach their target's owner by one variable-count all-to-all and every rank builds its rows of the graph;
    ring_lists: ring k-NN, lists all-gathered, graph stage replicated; replicated: the round-1 all-gather form."""
    import torch.multiprocessing as mp
    from oracle import oracle_np
    n, d, world = 300, 24, 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, d, split, out, mode), nprocs=world, join=True)
    X = clustered(n, d, nclust=6, seed=21)
    gp = {"eps": calibrate_eps(X, 6), "k": 6, "topk": 5, "p": 2.0, "sigma": None}
    ref = oracle_np.build(X, gp)
    rng = np.random.default_rng(5)
    want = []
    for _ in range(4):
        q = X[rng.integers(0, n)] + 0.02 * rng.standard_normal(d) / np.sqrt(d)
        for tau in (1.0, 0.62, 0.0):
            want.append(oracle_np.search(ref, q, tau))
    for rank in range(world):
        lam, res = out[rank]
        np.testing.assert_allclose(lam, ref["lambdas"], rtol=1e-12)
        for (hits, lq), (whits, wlq) in zip(res, want):
            assert [i for i, _ in hits] == [i for i, _ in whits]
            np.testing.assert_allclose([s for _, s in hits], [s for _, s in whits], rtol=1e-12)
            assert abs(lq - wlq) <= 1e-12 * abs(wlq)
    assert out[0][1] == out[1][1]          # every rank returns the same answer


@pytest.mark.parametrize("world,cuts", [(3, (70, 190)), (4, (60, 61, 200)), (6, (40, 95, 96, 180, 260))])
def test_symmetric_ring_with_three_and_four_ranks(world, cuts):
    """world 3: every pair is a whole pair (steps 1 .. 1); world 4: one whole-pair step and the split pair at distance 2
    (the lower rank takes the first half of the other block's column tiles, the higher rank its rows of the second
    half) -- uneven shards, one of a single row.  Lists, lambdas and searches equal the single-process oracle's."""
    import torch.multiprocessing as mp
    from oracle import oracle_np
    n, d = 300, 24
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, d, cuts, out, "ring"), nprocs=world, join=True)
    X = clustered(n, d, nclust=6, seed=21)
    gp = {"eps": calibrate_eps(X, 6), "k": 6, "topk": 5, "p": 2.0, "sigma": None}
    ref = oracle_np.build(X, gp)
    for rank in range(world):
        np.testing.assert_allclose(out[rank][0], ref["lambdas"], rtol=1e-12)
    assert all(out[r][1] == out[0][1] for r in range(world))


class RoundsEngine(OracleEngine):
    """The ring's second and third rounds on the CPU: one rank reports unproven rows after the first merge, another one
    overflowed bands after the second round -- every rank must then go round again (the rounds are collective), and the
    lists recomputed from the visiting blocks are the same exact lists."""

    def __init__(self, gp, rank, flag_rank, over_rank):
        super().__init__(gp)
        self.rank, self.flag_rank, self.over_rank = rank, flag_rank, over_rank
        self.merges, self.band_blocks, self.exact_blocks = 0, [], []

    def knn_merge(self, nmax, final=False):
        import torch
        k = self.prm["k"]
        idx = np.full((self.n, k), -1, dtype=np.int32)
        dist = np.zeros((self.n, k))
        gy = np.zeros((self.n, k))
        cnt = np.zeros(self.n, dtype=np.int32)
        for i in range(self.n):      # (a block seen in several rounds contributes the same entries again: a set)
            allc = sorted(set(c for b in self.parts for c in self.parts[b][i]))[:k]
            for t, (kk, j, dd, gg) in enumerate(allc):
                idx[i, t], dist[i, t], gy[i, t] = j, dd, gg
            cnt[i] = len(allc)
        self._lists = tuple(torch.from_numpy(a) for a in (idx, dist, gy, cnt))
        self.merges += 1
        self.final = final
        return 3 if self.merges == 1 and self.rank == self.flag_rank else 0

    def knn_block_band(self, h, b, row_goff, col_goff):
        self.band_blocks.append(b)
        self.knn_block(h, b, row_goff, col_goff)
        return 0

    def overflowed_rows(self):
        return 2 if self.rank == self.over_rank else 0

    def knn_block_exact(self, h, b, row_goff, col_goff):
        self.exact_blocks.append(b)
        self.knn_block(h, b, row_goff, col_goff)


def _rounds_worker(rank, world, port, n, d, cuts, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pyarrowspace_amd.dist import ShardedIndex
        X = clustered(n, d, nclust=6, seed=21)
        gp = {"eps": calibrate_eps(X, 6), "k": 6, "topk": 5, "p": 2.0, "sigma": None}
        bounds = sorted([0, n] + list(cuts))
        shard = torch.from_numpy(X[bounds[rank] : bounds[rank + 1]].copy())
        e = RoundsEngine(gp, rank, flag_rank=0, over_rank=world - 1)
        index = ShardedIndex.build(gp, shard, dist, engine=e)
        # every rank saw every block in the second and in the third round; only the overflowed rank merged once more
        assert sorted(e.band_blocks) == list(range(world)) and sorted(e.exact_blocks) == list(range(world))
        assert e.merges == (3 if rank == world - 1 else 2) and e.final == (rank == world - 1)
        assert index.ring_flagged == (3 if rank == 0 else 0) and index.ring_overflowed == (2 if rank == world - 1 else 0)
        q = X[17] + 0.01
        out[rank] = (index.lambdas().copy(), [index.search(q, tau) for tau in (1.0, 0.62)])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,cuts", [(2, (120,)), (3, (70, 190)), (4, (60, 61, 200))])
def test_second_and_third_ring_rounds_are_collective(world, cuts):
    """Unproven rows on one rank, overflowed bands on another: all ranks send their shards round a second and a third
    time (ShardedIndex._ring_knn), the rank without flags just passes blocks on; same index as the single process."""
    import torch.multiprocessing as mp
    from oracle import oracle_np
    n, d = 300, 24
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rounds_worker, args=(world, _free_port(), n, d, cuts, out), nprocs=world, join=True)
    X = clustered(n, d, nclust=6, seed=21)
    gp = {"eps": calibrate_eps(X, 6), "k": 6, "topk": 5, "p": 2.0, "sigma": None}
    ref = oracle_np.build(X, gp)
    for rank in range(world):
        np.testing.assert_allclose(out[rank][0], ref["lambdas"], rtol=1e-12)
    assert all(out[r][1] == out[0][1] for r in range(world))
    want = [oracle_np.search(ref, X[17] + 0.01, tau) for tau in (1.0, 0.62)]
    for hits, (whits, _) in zip(out[0][1], want):
        assert [i for i, _ in hits] == [i for i, _ in whits]


def test_shard_bounds():
    from pyarrowspace_amd.dist import shard_bounds
    assert shard_bounds(10, 4) == [0, 3, 6, 8, 10]
    assert shard_bounds(3, 8) == [0, 1, 2, 3, 3, 3, 3, 3, 3]
    assert shard_bounds(1_000_000, 8)[-1] == 1_000_000


def test_record_layouts_match_the_c_structs():
    from pyarrowspace_amd import _lib
    from pyarrowspace_amd.dist import HIT_REC_F64, KNN_REC_F64
    import ctypes
    assert ctypes.sizeof(_lib.KnnRec) == 8 * KNN_REC_F64
    assert ctypes.sizeof(_lib.HitRec) == 8 * HIT_REC_F64


def test_escalation_steps():
    """pyarrowspace_amd.dist.next_mode: fp64 restart, threshold repair, list path -- and termination."""
    from pyarrowspace_amd.dist import next_mode
    assert next_mode(0, False, 0) is None
    assert next_mode(0, False, 1) == 4            # crowded neighbourhood: repair from the kept dots
    assert next_mode(4, False, 0) is None
    assert next_mode(4, False, 1) == 2            # the repair overflowed as well (mass ties): list path
    assert next_mode(0, False, 2) == 2            # scorer buffer overflowed
    assert next_mode(0, False, 3) == 4            # the scorer's bit means nothing while the neighbourhood is truncated
    assert next_mode(4, False, 2) == 2
    assert next_mode(2, False, 3) is None         # nothing beyond the list path in fp32 ...
    assert next_mode(2, True, 0) == 3             # ... except fp64
    assert next_mode(0, True, 0) == 1             # not provably exact in fp32
    assert next_mode(4, True, 0) == 1             # fp32 dots are no use to the fp64 pass
    assert next_mode(4, True, 2) == 3             # fp64 and the list path in one step
    assert next_mode(1, True, 0) is None
    assert next_mode(1, False, 1) == 5
    assert next_mode(5, False, 1) == 3
    assert next_mode(3, True, 3) is None
    # every chain ends within four passes whatever the flags
    import itertools
    for flags in itertools.product([(False, 0), (False, 1), (False, 2), (False, 3), (True, 0), (True, 1), (True, 2), (True, 3)], repeat=4):
        m, steps, seen = 0, 0, []
        for f in flags:
            seen.append(m)
            m = next_mode(m, *f)
            steps += 1
            if m is None:
                break
            assert m not in seen, (flags, seen, m)
        else:
            assert next_mode(m, True, 3) is None, (flags, m)


def test_sharded_search_follows_the_escalation_ladder():
    """ShardedIndex.search with an engine that raises flags on a script: crowded neighbourhood -> repair (mode 4),
    repair overflows -> list path (2), not provably exact -> fp64 (3); the answer of the last pass is returned and
    the loop ends within four passes."""
    import torch
    from pyarrowspace_amd.dist import ShardedIndex

    class Scripted(OracleEngine):
        script = []

        def __init__(self, gp):
            super().__init__(gp)
            self.modes = []

        def set_mode(self, mode):
            self.modes.append(mode)
            super().set_mode(mode)

        def query_finish(self, hits_all):
            hits, lq, zero, _, _ = super().query_finish(hits_all)
            inexact, overflow = self.script[len(self.modes) - 1] if len(self.modes) <= len(self.script) else (False, 0)
            return hits, lq, zero, inexact, overflow

    X = np.random.default_rng(0).standard_normal((120, 16))
    gp = {"eps": 6.0, "k": 4, "topk": 3, "p": 2.0, "sigma": None}
    plain = ShardedIndex.build(gp, torch.from_numpy(X), engine=OracleEngine(gp))
    want = plain.search(X[3] * 1.01, 0.62)
    for script, modes in (([(False, 0)], [0]),
                          ([(False, 1), (False, 0)], [0, 4]),
                          ([(False, 3), (False, 1), (False, 0)], [0, 4, 2]),
                          ([(False, 1), (False, 1), (True, 0), (False, 0)], [0, 4, 2, 3]),
                          ([(False, 3), (True, 2), (False, 0)], [0, 4, 3]),
                          ([(True, 0), (False, 1), (False, 2)], [0, 1, 5, 3])):
        eng = Scripted(gp)
        eng.script = script
        index = ShardedIndex.build(gp, torch.from_numpy(X), engine=eng)
        assert index.search(X[3] * 1.01, 0.62) == want
        assert eng.modes == modes, (script, eng.modes)
