"""Row-sharded multi-GPU build and search: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI) for the exchange steps, the C ABI for every kernel.

Build   (DESIGN.md section 6): all-gather of the item shards (each rank ends with the full
        fp32 item matrix in HBM) -> every rank runs the fused X.X^T k-NN kernel for ITS rows
        against all columns -> all-gather of the exact k-NN lists -> the O(N k) graph /
        Laplacian / lambda stage is computed redundantly on every rank (milliseconds).
Search: every rank scans its own rows, the k nearest-neighbour records and the top-k hit
        records (fixed-size structs of include/arrowspace_hip.h) are all-gathered and merged
        identically on every rank; two collectives of a few KB per query.

The host logic is engine-agnostic: `HipEngine` (this file) drives libarrowspace_hip.so;
tests inject a CPU engine to exercise the sharding logic under the gloo backend.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

KNN_REC_F64 = 6   # as_knn_rec = {int64 idx; double key, dist, gy, deg, ny}
HIT_REC_F64 = 2   # as_hit_rec = {int64 idx; double score}


def shard_bounds(n: int, world: int) -> list[int]:
    """Contiguous row ranges: rank r owns [b[r], b[r+1])."""
    base, rem = divmod(n, world)
    b = [0]
    for r in range(world):
        b.append(b[-1] + base + (1 if r < rem else 0))
    return b


def next_mode(mode: int, inexact: bool, overflow: int):
    """Escalation of one sharded search (every rank sees the same merged flags, so every rank takes the
    same step).  mode bits as in as_query_set_exact: 1 = fp64 scans, 2 = wavefront-list selection,
    4 = repair the k-NN candidates from the kept dots.  Returns None when the answer stands (or nothing
    stronger exists: fp64 + list selection).  Every step is strictly stronger, so a search takes at most
    four passes: 0 -> 4 -> 2 -> 3, or 0 -> 1 -> 5 -> 3."""
    if inexact and not mode & 1:
        # fp64 from scratch (the fp32 dots cannot be reused); a buffer that overflowed will overflow again,
        # so take the list path along with it
        return 3 if overflow or mode & 2 else 1
    if overflow & 1 and not mode & 6:
        # more than CAND_CAP rows inside eps: threshold repair, no rescan (the scorer ran on a truncated
        # neighbourhood, so its own overflow bit means nothing yet)
        return mode | 4
    if overflow and not mode & 2:
        return (mode | 2) & ~4                    # scorer buffer, or the repair itself overflowed: list path
    return None


class HipEngine:
    """Thin driver of the staged C ABI on this rank's GPU (torch tensors carry the pointers)."""

    def __init__(self, graph_params):
        import torch

        from . import _lib, _parse_graph_params

        self.torch = torch
        self._lib = _lib
        self.L = _lib.load()
        self.gp, self.op = _parse_graph_params(graph_params)
        self.op.device = torch.cuda.current_device()
        self.sp = C.c_void_p()
        self.gr = C.c_void_p()
        self.q = C.c_void_p()
        self._keep = []

    def _check(self, st):
        if st:
            from . import _raise
            _raise(st)

    # ---- build
    def create_space(self, X):
        torch = self.torch
        assert X.is_cuda and X.is_contiguous() and X.dtype in (torch.float32, torch.float64)
        dt = self._lib.DTYPE_F32 if X.dtype == torch.float32 else self._lib.DTYPE_F64
        torch.cuda.current_stream().synchronize()
        self._check(self.L.as_space_create_dev(C.c_void_p(X.data_ptr()), dt, X.shape[0], X.shape[1], X.shape[1],
                                               C.byref(self.op), C.byref(self.sp)))
        self.n, self.d = int(X.shape[0]), int(X.shape[1])

    def knn_rows(self, r0, r1):
        torch = self.torch
        k, rows = int(self.gp.k), r1 - r0
        dev = torch.device("cuda", self.op.device)
        idx = torch.full((max(rows, 1), k), -1, dtype=torch.int32, device=dev)
        key = torch.zeros((max(rows, 1), k), dtype=torch.float64, device=dev)
        dist = torch.zeros_like(key)
        gy = torch.zeros_like(key)
        cnt = torch.zeros((max(rows, 1),), dtype=torch.int32, device=dev)
        torch.cuda.current_stream().synchronize()
        if rows > 0:
            self._check(self.L.as_knn_rows(self.sp, C.byref(self.gp), r0, r1, C.c_void_p(idx.data_ptr()),
                                           C.c_void_p(key.data_ptr()), C.c_void_p(dist.data_ptr()),
                                           C.c_void_p(gy.data_ptr()), C.c_void_p(cnt.data_ptr())))
        return idx[:rows], dist[:rows], gy[:rows], cnt[:rows]

    def graph_from_knn(self, idx, dist, gy, cnt):
        self.torch.cuda.current_stream().synchronize()
        self._check(self.L.as_graph_from_knn(self.sp, C.byref(self.gp), C.c_void_p(idx.data_ptr()),
                                             C.c_void_p(dist.data_ptr()), C.c_void_p(gy.data_ptr()),
                                             C.c_void_p(cnt.data_ptr()), C.byref(self.gr)))

    # ---- search
    def query_open(self):
        torch = self.torch
        self._check(self.L.as_query_create(self.sp, self.gr, C.byref(self.q)))
        self.k = int(self.L.as_query_knn_capacity(self.q))
        self.hcap = int(self.L.as_query_hit_capacity(self.q))
        dev = torch.device("cuda", self.op.device)
        self.knn_local = torch.zeros((self.k, KNN_REC_F64), dtype=torch.float64, device=dev)
        self.hits_local = torch.zeros((self.hcap, HIT_REC_F64), dtype=torch.float64, device=dev)
        self._check(self.L.as_query_bind_records(self.q, C.c_void_p(self.knn_local.data_ptr()),
                                                 C.c_void_p(self.hits_local.data_ptr())))
        # run the query kernels on torch's current stream: RCCL collectives order against it
        self.L.as_query_set_stream(self.q, C.c_void_p(torch.cuda.current_stream().cuda_stream))
        self.topk = self.hcap - 1
        self._idx = np.empty(max(self.topk, 1), dtype=np.int64)
        self._sc = np.empty(max(self.topk, 1), dtype=np.float64)

    def set_mode(self, mode):
        self.L.as_query_set_exact(self.q, int(mode))

    def query_scan(self, q, r0, r1):
        self._q = np.ascontiguousarray(q, dtype=np.float64)
        self._check(self.L.as_query_scan(self.q, self._q.ctypes.data_as(C.c_void_p), self._q.shape[0], r0, r1))

    def query_lambda(self, knn_all):
        self._check(self.L.as_query_lambda(self.q, C.c_void_p(knn_all.data_ptr()), knn_all.shape[0]))

    def query_score(self, tau):
        self._check(self.L.as_query_score(self.q, float(tau)))

    def query_finish(self, hits_all):
        ln, lq = C.c_int64(0), C.c_double(0.0)
        st = self.L.as_query_finish(self.q, C.c_void_p(hits_all.data_ptr()), hits_all.shape[0],
                                    self._idx.ctypes.data_as(C.c_void_p), self._sc.ctypes.data_as(C.c_void_p),
                                    C.byref(ln), C.byref(lq))
        ki, si = C.c_int32(0), C.c_int32(0)
        self.L.as_query_flags(self.q, C.byref(ki), C.byref(si))
        inexact = bool((ki.value & 1) or (si.value & 1))
        overflow = (1 if ki.value & 2 else 0) | (2 if si.value & 2 else 0)   # bit0: k-NN buffer, bit1: scorer buffer
        if st not in (self._lib.AS_OK, self._lib.AS_EZEROLAMBDA):
            self._check(st)
        hits = [(int(self._idx[t]), float(self._sc[t])) for t in range(ln.value)]
        return hits, float(lq.value), st == self._lib.AS_EZEROLAMBDA, inexact, overflow

    def lambdas(self):
        out = np.empty(self.n, dtype=np.float64)
        self._check(self.L.as_lambdas(self.sp, out.ctypes.data_as(C.c_void_p)))
        return out

    def stats(self):
        out = np.zeros(8, dtype=np.float64)
        self.L.as_build_stats(self.gr, out.ctypes.data_as(C.c_void_p), 8)
        keys = ("ingest_s", "knn_mfma_s", "refine_s", "fallback_s", "graph_s", "total_s", "fallback_rows", "mfma_flops")
        return dict(zip(keys, out.tolist()))

    def tau0(self):
        return float(self.L.as_graph_tau0(self.gr))

    def scan_us(self):
        out = np.zeros(3, dtype=np.float64)
        self.L.as_query_stats(self.q, out.ctypes.data_as(C.c_void_p), 3)
        return float(out[0])

    def close(self):
        if self.q:
            self.L.as_query_free(self.q)
            self.q = C.c_void_p()
        if self.gr:
            self.L.as_free_graph(self.gr)
            self.gr = C.c_void_p()
        if self.sp:
            self.L.as_free_space(self.sp)
            self.sp = C.c_void_p()


class ShardedIndex:
    """Host logic of the row-sharded index; identical results on every rank."""

    def __init__(self):
        self.engine = None
        self._gbuf = {}
        self.stream = None
        self.force_collectives = False

    # ---- collectives
    def _collective(self):
        return self.world > 1 or self.force_collectives

    def _gather_rows(self, t, counts):
        """t: this rank's [rows_r, ...] tensor -> concatenation over ranks, [sum(counts), ...]."""
        torch = self.torch
        world = len(counts)
        if not self._collective():
            return t
        mx = max(max(counts), 1)
        if t.is_cuda and all(c == mx for c in counts):
            # RCCL all-gather straight into the final buffer (equal shards: no padding, no copies)
            out = torch.empty((world * mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            self.dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
            return out
        pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        if t.is_cuda:
            buf = torch.empty((world * mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            self.dist.all_gather_into_tensor(buf, pad, group=self.group)
            parts = [buf[r * mx : r * mx + counts[r]] for r in range(world)]
        else:
            full = [torch.empty_like(pad) for _ in range(world)]
            self.dist.all_gather(full, pad, group=self.group)
            parts = [full[r][: counts[r]] for r in range(world)]
        return torch.cat(parts, dim=0)

    def _gather_fixed(self, t):
        torch = self.torch
        if not self._collective():
            return t
        if t.is_cuda:
            # one collective into a preallocated [world * rows, ...] buffer, no per-call allocation
            key = (t.data_ptr(), tuple(t.shape))
            out = self._gbuf.get(key)
            if out is None:
                out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
                self._gbuf[key] = out
            self.dist.all_gather_into_tensor(out, t, group=self.group)
            return out
        parts = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t, group=self.group)
        return torch.cat(parts, dim=0)

    @classmethod
    def build(cls, graph_params, X_shard, dist=None, group=None, engine=None, force_collectives=False):
        """X_shard: this rank's contiguous block of rows (torch tensor on this rank's device,
        fp32 or fp64).  Ranks hold consecutive blocks in rank order."""
        import contextlib

        import torch

        self = cls()
        self.torch, self.dist, self.group = torch, dist, group
        self.world = dist.get_world_size(group) if dist is not None else 1
        self.rank = dist.get_rank(group) if dist is not None else 0
        self.force_collectives = bool(force_collectives) and dist is not None
        self.engine = engine if engine is not None else HipEngine(graph_params)
        # CUDA: kernels and RCCL collectives are ordered on ONE dedicated, non-default stream
        # (the legacy default stream has handle 0 and cannot be handed to the library)
        self.stream = torch.cuda.Stream() if X_shard.is_cuda else None
        ctx = torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
        with ctx:
            rows = int(X_shard.shape[0])
            if self.world > 1:
                c = torch.tensor([rows], dtype=torch.int64, device=X_shard.device)
                cs = [torch.zeros_like(c) for _ in range(self.world)]
                dist.all_gather(cs, c, group=group)
                counts = [int(v.item()) for v in cs]
            else:
                counts = [rows]
            self.counts = counts
            self.bounds = [0]
            for v in counts:
                self.bounds.append(self.bounds[-1] + v)
            self.n = self.bounds[-1]
            self.r0, self.r1 = self.bounds[self.rank], self.bounds[self.rank + 1]
            X_full = self._gather_rows(X_shard.contiguous(), counts).contiguous()
            self._sync()
            self.engine.create_space(X_full)
            del X_full
            idx, dst, gy, cnt = self.engine.knn_rows(self.r0, self.r1)
            idx = self._gather_rows(idx, counts).contiguous()
            dst = self._gather_rows(dst, counts).contiguous()
            gy = self._gather_rows(gy, counts).contiguous()
            cnt = self._gather_rows(cnt, counts).contiguous()
            self._sync()
            self.engine.graph_from_knn(idx, dst, gy, cnt)
            self.engine.query_open()
        return self

    def _sync(self):
        if self.stream is not None:
            self.stream.synchronize()

    def search(self, q, tau):
        """Same contract as ArrowSpace.search (src/lib.rs:132-174); every rank passes the same q."""
        import contextlib

        from . import PanicException

        e = self.engine
        ctx = self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()
        mode = 0
        with ctx:
            for _ in range(8):
                e.set_mode(mode)
                e.query_scan(q, self.r0, self.r1)
                knn_all = self._gather_fixed(e.knn_local)
                e.query_lambda(knn_all)
                e.query_score(tau)
                hits_all = self._gather_fixed(e.hits_local)
                hits, lq, zero, inexact, overflow = e.query_finish(hits_all)
                mode = next_mode(mode, inexact, overflow)
                if mode is None:
                    break
            else:   # next_mode is strictly increasing: unreachable, and never a silently wrong answer
                raise RuntimeError("sharded search did not settle on an exact answer")
        self.last_lambda_q = lq
        if zero:
            raise PanicException("The lambdas are zero, check the magnitude of items and eps.")
        return hits

    def lambdas(self):
        return self.engine.lambdas()

    def last_scan_us(self):
        return self.engine.scan_us()

    def build_stats(self):
        return getattr(self.engine, "stats", lambda: {})()

    def close(self):
        if self.engine is not None:
            self.engine.close()
            self.engine = None
