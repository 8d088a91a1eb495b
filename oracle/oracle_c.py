"""ctypes wrapper over oracle/libarrowspace_oracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libarrowspace_oracle.so")
_lib = None

METRICS = {"l2": 0, "cosine": 1}
KERNELS = {"gaussian": 0, "rational": 1}


def build_lib(force: bool = False) -> str:
    src = os.path.join(_HERE, "arrowspace_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build_lib()
        L = C.CDLL(_SO)
        p64 = C.POINTER(C.c_double)
        pi64 = C.POINTER(C.c_int64)
        L.aso_build.restype = C.c_void_p
        L.aso_build.argtypes = [p64, C.c_int64, C.c_int64, C.c_double, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_int]
        L.aso_build_feature.restype = C.c_void_p
        L.aso_build_feature.argtypes = L.aso_build.argtypes
        L.aso_free.argtypes = [C.c_void_p]
        L.aso_query_lambda.restype = C.c_double
        L.aso_query_lambda.argtypes = [C.c_void_p, p64]
        L.aso_search.restype = C.c_int64
        L.aso_search.argtypes = [C.c_void_p, p64, C.c_double, C.c_int64, pi64, p64, p64]
        L.aso_search_fused.restype = C.c_int64
        L.aso_search_fused.argtypes = L.aso_search.argtypes
        L.aso_search_with_lambda.restype = C.c_int64
        L.aso_search_with_lambda.argtypes = [C.c_void_p, p64, C.c_double, C.c_double, C.c_int64, pi64, p64]
        L.aso_scores.argtypes = [C.c_void_p, p64, C.c_double, C.c_double, p64]
        for name in ("aso_n", "aso_d", "aso_nnz", "aso_nnodes"):
            getattr(L, name).restype = C.c_int64
            getattr(L, name).argtypes = [C.c_void_p]
        L.aso_tau0.restype = C.c_double
        L.aso_tau0.argtypes = [C.c_void_p]
        for name in ("aso_dist", "aso_gy", "aso_w", "aso_lap", "aso_deg", "aso_E", "aso_G", "aso_lambdas", "aso_norms"):
            getattr(L, name).restype = p64
            getattr(L, name).argtypes = [C.c_void_p]
        for name in ("aso_indptr", "aso_indices", "aso_knn_idx", "aso_knn_cnt"):
            getattr(L, name).restype = pi64
            getattr(L, name).argtypes = [C.c_void_p]
        L.aso_threads.restype = C.c_int
        L.aso_set_threads.argtypes = [C.c_int]
        # a container's CPU quota, not the host's core count: 128 OpenMP threads on a quota of 16 CPUs spend their time
        # throttled, and "cores" in the bench line would overstate what ran (OMP_NUM_THREADS set by the caller wins)
        quota = _cpu_quota()
        if quota and "OMP_NUM_THREADS" not in os.environ and quota < int(L.aso_threads()):
            L.aso_set_threads(quota)
        L.aso_from_parts.restype = C.c_void_p
        L.aso_from_parts.argtypes = [p64, C.c_int64, C.c_int64, C.c_double, C.c_int64, C.c_double, C.c_double, C.c_int, C.c_int, p64, p64, C.c_double]
        L.aso_free_parts.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _cpu_quota():
    """CPUs this process may use: the cgroup quota (v2 cpu.max, v1 cfs_quota / cfs_period) and the affinity mask."""
    q = None
    try:
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if a != "max":
            q = max(1, int(int(a) / int(b)))
    except (OSError, ValueError):
        try:
            a = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            b = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if a > 0 and b > 0:
                q = max(1, a // b)
        except (OSError, ValueError):
            pass
    try:
        aff = len(os.sched_getaffinity(0))
        q = aff if q is None else min(q, aff)
    except (AttributeError, OSError):
        pass
    return q


def _p64(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class ZeroLambda(Exception):
    pass


class OracleIndex:
    """fp64 CPU index built by the C restatement; mirrors oracle_np.build()'s dict."""

    def __init__(self, X, graph_params: dict):
        from .oracle_np import resolve_params

        X = np.ascontiguousarray(X, dtype=np.float64)
        if X.ndim != 2 or X.shape[0] == 0 or X.shape[1] == 0:
            raise ValueError("items must be non-empty 2D array")
        self.prm = resolve_params(graph_params)
        self.X = X
        L = lib()
        fn = L.aso_build_feature if self.prm["lambda_mode"] == 1 else L.aso_build
        self._h = fn(_p64(X), X.shape[0], X.shape[1], self.prm["eps"], self.prm["k"], self.prm["p"],
                     self.prm["sigma"], self.prm["metric"], self.prm["kernel"])
        if not self._h:
            raise ValueError("aso_build failed")
        nitems = X.shape[0]
        n = L.aso_nnodes(self._h)   # graph nodes: the items, or the D features in feature mode
        nnz = L.aso_nnz(self._h)
        k = self.prm["k"]

        def arr(ptr, m, dt):
            return np.ctypeslib.as_array(ptr, shape=(max(m, 1),))[:m].astype(dt, copy=True)

        self.nnodes = n
        self.n = arr(L.aso_norms(self._h), nitems, np.float64)
        self.indptr = arr(L.aso_indptr(self._h), n + 1, np.int64)
        self.indices = arr(L.aso_indices(self._h), nnz, np.int64)
        self.dist = arr(L.aso_dist(self._h), nnz, np.float64)
        self.gy = arr(L.aso_gy(self._h), nnz, np.float64)
        self.w = arr(L.aso_w(self._h), nnz, np.float64)
        self.lap = arr(L.aso_lap(self._h), nnz, np.float64)
        self.deg = arr(L.aso_deg(self._h), n, np.float64)
        self.E = arr(L.aso_E(self._h), nitems, np.float64)
        self.G = arr(L.aso_G(self._h), nitems, np.float64)
        self.lambdas = arr(L.aso_lambdas(self._h), nitems, np.float64)
        self.knn_idx = arr(L.aso_knn_idx(self._h), n * k, np.int64).reshape(n, k)
        self.knn_cnt = arr(L.aso_knn_cnt(self._h), n, np.int64)
        self.tau0 = L.aso_tau0(self._h)

    def __del__(self):
        if getattr(self, "_h", None):
            lib().aso_free(self._h)
            self._h = None

    def query_lambda(self, q) -> float:
        q = np.ascontiguousarray(q, dtype=np.float64)
        return lib().aso_query_lambda(self._h, _p64(q))

    def scores(self, q, tau, lambda_q):
        q = np.ascontiguousarray(q, dtype=np.float64)
        out = np.empty(self.X.shape[0])
        lib().aso_scores(self._h, _p64(q), tau, lambda_q, _p64(out))
        return out

    def search(self, q, tau, topk=None, fused=False):
        """fused=True: the one-pass form (aso_search_fused), same results."""
        q = np.ascontiguousarray(q, dtype=np.float64)
        if q.ndim != 1 or q.shape[0] != self.X.shape[1]:
            raise ValueError(f"query length {q.shape[0]} must match nfeatures {self.X.shape[1]}")
        topk = self.prm["topk"] if topk is None else topk
        kk = min(topk, self.X.shape[0])
        idx = np.empty(kk, dtype=np.int64)
        sc = np.empty(kk)
        lq = C.c_double(0.0)
        fn = lib().aso_search_fused if fused else lib().aso_search
        m = fn(self._h, _p64(q), tau, kk, idx.ctypes.data_as(C.POINTER(C.c_int64)), _p64(sc), C.byref(lq))
        if m < 0:
            raise ZeroLambda("The lambdas are zero, check the magnitude of items and eps.")
        return [(int(idx[t]), float(sc[t])) for t in range(m)], lq.value


class OracleSearchOnly:
    """CPU scorer over precomputed lambdas/degrees (bench.py cpu_baseline leg)."""

    def __init__(self, X, graph_params, deg, lambdas, tau0):
        from .oracle_np import resolve_params

        self.X = np.ascontiguousarray(X, dtype=np.float64)   # the library keeps its own, first-touched copy
        self.prm = resolve_params(graph_params)
        self._deg = np.ascontiguousarray(deg, dtype=np.float64)
        self._lam = np.ascontiguousarray(lambdas, dtype=np.float64)
        self._h = lib().aso_from_parts(_p64(self.X), self.X.shape[0], self.X.shape[1], self.prm["eps"], self.prm["k"],
                                       self.prm["p"], self.prm["sigma"], self.prm["metric"], self.prm["kernel"],
                                       _p64(self._deg), _p64(self._lam), float(tau0))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().aso_free_parts(self._h)
            self._h = None

    search = OracleIndex.search
    scores = OracleIndex.scores
    query_lambda = OracleIndex.query_lambda


def threads() -> int:
    return int(lib().aso_threads())
