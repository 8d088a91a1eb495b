"""pyarrowspace_amd -- MI355X-native drop-in for the `arrowspace` Python module.

Host-side mirror of the reference's PyO3 surface (/root/reference/src/lib.rs:379-386):
`ArrowSpaceBuilder`, `ArrowSpace`, `GraphLaplacian`, `set_debug`.  Same names, same
argument order and meaning, same error behaviour; the arithmetic runs in hand-written
HIP kernels behind the C ABI of include/arrowspace_hip.h (libarrowspace_hip.so).
There is no CPU fallback: importing this package without the built library fails.
"""
from __future__ import annotations

import ctypes as C
import threading
import os

import numpy as np

from . import _lib
from ._lib import GraphParams, Opts

__all__ = ["ArrowSpaceBuilder", "ArrowSpace", "GraphLaplacian", "set_debug", "PanicException"]
__version__ = "0.1.0"

_L = _lib.load()


class PanicException(BaseException):
    """Mirror of pyo3_runtime.PanicException (a BaseException): raised where the
    reference panics, i.e. `assert_ne!(lambda_q, 0.0)` at src/lib.rs:156-159."""


def _raise(status: int):
    msg = _lib.last_error()
    if status == _lib.AS_EZEROLAMBDA:
        raise PanicException(msg or "The lambdas are zero, check the magnitude of items and eps.")
    if status in (_lib.AS_EINVAL, _lib.AS_EUNSUPPORTED):
        raise ValueError(msg)
    if status == _lib.AS_ENOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def set_debug(enabled: bool) -> None:
    """src/helpers.rs:12-14: process-global flag; messages go to stderr as `[pyarrowspace] ...`."""
    _L.as_set_debug(1 if enabled else 0)


if os.environ.get("ARROWSPACE_DEBUG", "") not in ("", "0"):
    set_debug(True)


def enable_search_stats(enabled: bool) -> None:
    """Extension: record HIP events around the scan kernel of every search (off by default)."""
    _L.as_enable_search_stats(1 if enabled else 0)


NORTH_STAR_MODE = {"metric": "l2", "kernel": "gaussian", "lambda_mode": "item"}        # BASELINE.json north_star
REFERENCE_MODE = {"metric": "cosine", "kernel": "rational", "lambda_mode": "item"}      # GRAPH_VARIABLES.md:7-10


def _parse_graph_params(graph_params, mode=None) -> tuple[GraphParams, Opts]:
    """src/helpers.rs:48-76.  eps,k,topk,p required; sigma missing/None -> eps*0.5.
    Keys the reference ignores select the documented variants: 'metric' in
    {'l2','cosine'}, 'kernel' in {'gaussian','rational'}, 'lambda_mode' in {'item','feature'}
    (DESIGN.md section 2).  Precedence: the dict's keys, then env ARROWSPACE_METRIC / _KERNEL / _LAMBDA_MODE, then
    `mode` -- the defaults of the module the caller imported (NORTH_STAR_MODE for pyarrowspace_amd and bench.py,
    REFERENCE_MODE for the reference-named module `arrowspace`, whose scripts' parameter sets are rectified-cosine
    distances)."""
    mode = NORTH_STAR_MODE if mode is None else mode
    gp, op = GraphParams(), Opts()
    op.device = int(os.environ.get("ARROWSPACE_DEVICE", "-1"))
    metric = os.environ.get("ARROWSPACE_METRIC", mode["metric"])
    kernel = os.environ.get("ARROWSPACE_KERNEL", mode["kernel"])
    lmode = os.environ.get("ARROWSPACE_LAMBDA_MODE", mode["lambda_mode"])
    if graph_params is None:
        # builder defaults (GRAPH_VARIABLES.md:15: eps~1e-3, k~6, p=2, sigma:=eps)
        gp.eps, gp.k, gp.topk, gp.p, gp.sigma, gp.has_sigma = 1e-3, 6, 3, 2.0, 1e-3, 1
    else:
        if not isinstance(graph_params, dict):
            raise TypeError("graph_params must be a dict or None")
        for key in ("eps", "k", "topk", "p"):
            if key not in graph_params:
                raise ValueError(f"graph_params['{key}'] is required")
        gp.eps = float(graph_params["eps"])
        for key in ("k", "topk"):
            v = graph_params[key]
            if isinstance(v, bool) or not isinstance(v, (int, np.integer)) or v < 0:
                raise TypeError(f"graph_params['{key}'] must be a non-negative integer")
        gp.k, gp.topk = int(graph_params["k"]), int(graph_params["topk"])
        gp.p = float(graph_params["p"])
        sigma = graph_params.get("sigma", None)
        if sigma is None:
            gp.sigma, gp.has_sigma = 0.0, 0
        else:
            gp.sigma, gp.has_sigma = float(sigma), 1
        metric = graph_params.get("metric", metric)
        kernel = graph_params.get("kernel", kernel)
        lmode = graph_params.get("lambda_mode", lmode)
        op.force_exact = 1 if graph_params.get("force_exact", False) else 0
        op.keep_f64 = 1 if graph_params.get("keep_f64", False) else 0
        # test hook: start searches on a fallback path (bit0 fp64, bit1 wavefront-list selection)
        op.search_mode = int(graph_params.get("_search_mode", 0))
    if metric not in _lib.METRICS:
        raise ValueError(f"unknown metric {metric!r}; expected one of {sorted(_lib.METRICS)}")
    if kernel not in _lib.KERNELS:
        raise ValueError(f"unknown kernel {kernel!r}; expected one of {sorted(_lib.KERNELS)}")
    if lmode not in _lib.LAMBDA_MODES:
        raise ValueError(f"unknown lambda_mode {lmode!r}; expected one of {sorted(_lib.LAMBDA_MODES)}")
    op.metric, op.kernel, op.lambda_mode = _lib.METRICS[metric], _lib.KERNELS[kernel], _lib.LAMBDA_MODES[lmode]
    return gp, op


class GraphLaplacian:
    """Opaque handle of the normalised graph Laplacian (src/lib.rs:26-62)."""

    def __new__(cls, *a, **k):
        raise ValueError("GraphLaplacian cannot be constructed directly; use ArrowSpaceBuilder.build_with_graph")

    @classmethod
    def _wrap(cls, handle):
        self = object.__new__(cls)
        self._h = handle
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _L is not None:      # _L is None once the interpreter tears the module down
            _L.as_free_graph(h)
            self._h = None

    @property
    def nnodes(self) -> int:
        return int(_L.as_nnodes(self._h))

    def shape(self) -> tuple[int, int]:
        n = self.nnodes
        return (n, n)

    @property
    def graph_params(self) -> dict:
        """src/lib.rs:49-61: dict {eps,k,topk,p,sigma} with sigma resolved."""
        gp = GraphParams()
        st = _L.as_get_graph_params(self._h, C.byref(gp))
        if st:
            _raise(st)
        return {"eps": gp.eps, "k": int(gp.k), "topk": int(gp.topk), "p": gp.p, "sigma": gp.sigma}

    # ---- extensions (no reference counterpart) ----
    @property
    def tau0(self) -> float:
        return float(_L.as_graph_tau0(self._h))

    @property
    def lambda_mode(self) -> str:
        """'item': N-node item graph, normalised Laplacian.  'feature': the F x F feature-space
        Laplacian L = D - W of TAUMODE.md:8,12-27 (nnodes == nfeatures)."""
        return "feature" if _L.as_graph_lambda_mode(self._h) == 1 else "item"

    def degrees(self) -> np.ndarray:
        out = np.empty(self.nnodes, dtype=np.float64)
        st = _L.as_graph_degrees(self._h, out.ctypes.data_as(C.c_void_p))
        if st:
            _raise(st)
        return out

    def to_csr(self):
        """(indptr, indices, values), columns ascending: L = I - D^-1/2 W D^-1/2 of the item graph, or
        L = D - W of the feature graph (lambda_mode 'feature')."""
        n, nnz = self.nnodes, int(_L.as_graph_nnz(self._h))
        indptr = np.empty(n + 1, dtype=np.int64)
        indices = np.empty(nnz, dtype=np.int64)
        values = np.empty(nnz, dtype=np.float64)
        st = _L.as_graph_csr(self._h, indptr.ctypes.data_as(C.c_void_p), indices.ctypes.data_as(C.c_void_p),
                             values.ctypes.data_as(C.c_void_p))
        if st:
            _raise(st)
        return indptr, indices[: indptr[-1]], values[: indptr[-1]]

    def build_stats(self) -> dict:
        out = np.zeros(10, dtype=np.float64)
        _L.as_build_stats(self._h, out.ctypes.data_as(C.c_void_p), 10)
        keys = ("ingest_s", "knn_mfma_s", "refine_s", "fallback_s", "graph_s", "total_s", "fallback_rows", "mfma_flops",
                "unproven_rows", "band_rows")
        return dict(zip(keys, out.tolist()))


class ArrowSpace:
    """Items + per-item lambdas resident in HBM (src/lib.rs:64-263)."""

    def __new__(cls, *a, **k):
        raise ValueError("ArrowSpace cannot be constructed directly; use ArrowSpaceBuilder.build")

    @classmethod
    def _wrap(cls, handle):
        self = object.__new__(cls)
        self._h = handle
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and _L is not None:
            _L.as_free_space(h)
            self._h = None

    @property
    def nitems(self) -> int:
        return int(_L.as_nitems(self._h))

    @property
    def nfeatures(self) -> int:
        return int(_L.as_nfeatures(self._h))

    def get_item(self, idx: int):
        """src/lib.rs:100-120: (features ndarray[float64], lambda)."""
        if isinstance(idx, bool) or not isinstance(idx, (int, np.integer)):
            raise TypeError("idx must be an integer")
        if idx < 0:
            raise OverflowError("can't convert negative int to unsigned")
        out = np.empty(self.nfeatures, dtype=np.float64)
        lam = C.c_double(0.0)
        st = _L.as_get_item(self._h, int(idx), out.ctypes.data_as(C.c_void_p), C.byref(lam))
        if st:
            _raise(st)
        return out, float(lam.value)

    def lambdas(self) -> np.ndarray:
        out = np.empty(self.nitems, dtype=np.float64)
        st = _L.as_lambdas(self._h, out.ctypes.data_as(C.c_void_p))
        if st:
            _raise(st)
        return out

    @staticmethod
    def _query(item) -> np.ndarray:
        # PyReadonlyArray1<f64> + as_slice(): float64, 1-D, contiguous (src/lib.rs:139)
        if not isinstance(item, np.ndarray) or item.dtype != np.float64:
            raise TypeError("argument 'item': expected a 1-D numpy.ndarray of dtype float64")
        if item.ndim != 1:
            raise TypeError("argument 'item': expected a 1-D numpy.ndarray of dtype float64")
        if not item.flags.c_contiguous:
            raise ValueError("The given array is not contiguous")
        return item

    def search(self, item, gl: GraphLaplacian, tau: float):
        """src/lib.rs:132-174: list[(index, score)], length topk, score descending."""
        if not isinstance(gl, GraphLaplacian):
            raise TypeError("argument 'gl': expected GraphLaplacian")
        q = self._query(item)
        # per-(thread, space, graph) call state: output buffers and ctypes arguments are built once.  Per thread,
        # because ctypes drops the GIL during the call: the library serialises searches on a space, but shared
        # output buffers would be overwritten by the next thread's call before this one has read them.
        tls = self.__dict__.get("_tls")
        if tls is None:
            tls = self.__dict__.setdefault("_tls", threading.local())
        st_ = getattr(tls, "s", None)
        if st_ is None or st_[0] is not gl:
            topk = max(min(int(gl.graph_params["topk"]), self.nitems), 1)
            idx = (C.c_int64 * topk)()
            sc = (C.c_double * topk)()
            ln, lq = C.c_int64(0), C.c_double(0.0)
            st_ = tls.s = (gl, idx, sc, ln, lq, C.byref(ln), C.byref(lq))
        _, idx, sc, ln, lq, pln, plq = st_
        st = _L.as_search(self._h, gl._h, q.ctypes.data, q.shape[0], tau, idx, sc, pln, plq)
        if st:
            _raise(st)
        n = ln.value
        return list(zip(idx[:n], sc[:n]))

    def search_batch(self, items, gl: GraphLaplacian, tau: float):
        """Extension: B queries [B, D] -> list of B hit lists."""
        Q = np.ascontiguousarray(items, dtype=np.float64)
        if Q.ndim != 2:
            raise TypeError("items must be a 2-D float64 array")
        topk = min(int(gl.graph_params["topk"]), self.nitems)
        b = Q.shape[0]
        idx = np.empty((b, topk), dtype=np.int64)
        sc = np.empty((b, topk), dtype=np.float64)
        ln = np.zeros(b, dtype=np.int64)
        lq = np.zeros(b, dtype=np.float64)
        stt = np.zeros(b, dtype=np.int32)
        st = _L.as_search_batch(self._h, gl._h, Q.ctypes.data_as(C.c_void_p), b, Q.shape[1], float(tau),
                                idx.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p),
                                ln.ctypes.data_as(C.c_void_p), lq.ctypes.data_as(C.c_void_p),
                                stt.ctypes.data_as(C.c_void_p))
        if st:
            _raise(st)
        if (stt == _lib.AS_EZEROLAMBDA).any():
            raise PanicException("The lambdas are zero, check the magnitude of items and eps.")
        # (tolist + zip: element-wise int()/float() over the arrays cost 1.5 ms per 256 queries, a fifth of the GPU time)
        return [list(zip(ii, ss)) if l == topk else list(zip(ii[:l], ss[:l]))
                for ii, ss, l in zip(idx.tolist(), sc.tolist(), ln.tolist())]

    def last_search_stats(self) -> dict:
        """Extension: device microseconds of the last search (HIP events on its stream)."""
        out = np.zeros(3, dtype=np.float64)
        st = _L.as_last_search_stats(self._h, out.ctypes.data_as(C.c_void_p), 3)
        if st:
            _raise(st)
        return {"scan_us": out[0], "rest_us": out[1]}

    @property
    def unproven_searches(self) -> int:
        """Extension: searches whose answer failed its a-posteriori exactness check even on the fp64 path (ties inside
        fp64 rounding); the library also says so on stderr the first time.  0 in normal operation."""
        return int(_L.as_unproven_searches(self._h))

    def search_counters(self) -> dict:
        """Extension: what the single-query searches on this space cost so far -- reruns (a second pass over the items) by
        cause, zero-lambda results."""
        out = np.zeros(6, dtype=np.int64)
        st = _L.as_search_counters(self._h, out.ctypes.data_as(C.c_void_p), 6)
        if st:
            _raise(st)
        keys = ("searches", "zero_lambda", "rerun_knn_check", "rerun_overflow", "rerun_score_check", "searches_with_rerun")
        return dict(zip(keys, (int(v) for v in out)))

    @property
    def knn_pipe(self) -> str:
        """Extension: the matrix pipe the build's k-NN block ran on -- "int8" (two-digit image, the default where the items allow
        it), "bf16" (head + tail), "fp32" (ARROWSPACE_K2_FP32=1) or "none" (feature mode, loaded index)."""
        return {0: "fp32", 1: "bf16", 2: "int8"}.get(int(_L.as_space_knn_pipe(self._h)), "none")

    @property
    def last_scan_int8(self) -> bool:
        """Extension: the last single-query scan read the int8 two-digit image of the items (half the bytes) rather than fp32."""
        return bool(_L.as_last_scan_int8(self._h))

    @property
    def last_scan_operand(self) -> str:
        """Extension: what the last single-query scan read -- "fp32" items, their "int8" two-digit image, or "int8-high" (the
        image's high digits alone: the coarse scan, every candidate re-evaluated exactly)."""
        return {0: "fp32", 1: "int8", 2: "int8-high"}.get(int(_L.as_last_scan_int8(self._h)), "fp32")

    @property
    def last_batch_int8(self) -> bool:
        """Extension: the last batched pass of `search_batch` ran on the int8 images of items and queries (int8 matrix pipe)."""
        return bool(_L.as_last_batch_int8(self._h))

    @property
    def batch_dual_scans(self) -> int:
        """Extension: scans of `search_batch` that served two passes (64 queries) with one read of the items."""
        return int(_L.as_batch_dual_scans(self._h))

    @property
    def search_pool_size(self) -> int:
        """Extension: single-query workspaces the library holds for this space -- `search` is re-entrant across host threads
        (ctypes releases the GIL around the call), each concurrent call runs on a workspace and stream of its own."""
        return int(_L.as_search_pool_size(self._h))

    def gang_counters(self) -> list:
        """Extension: scans shared by concurrent `search` callers -- [launched with 1 member, 2, 3, 4] (gang scans: host threads whose
        searches arrive together are served by one pass over the items)."""
        out = np.zeros(4, dtype=np.int64)
        st = _L.as_gang_counters(self._h, out.ctypes.data_as(C.c_void_p), 4)
        if st:
            _raise(st)
        return [int(v) for v in out]

    def gang_skips(self) -> dict:
        """Extension: host-prepared single-query scans that did not take the shared path, by first reason."""
        out = np.zeros(10, dtype=np.int64)
        st = _L.as_gang_counters(self._h, out.ctypes.data_as(C.c_void_p), 10)
        if st:
            _raise(st)
        keys = ("no_concurrent_callers", "not_fused_tail", "not_coarse", "state_to_reset", "timed", "row_range")
        return dict(zip(keys, (int(v) for v in out[4:])))

    def save(self, gl: GraphLaplacian, path: str) -> None:
        """Extension: write the built index (items, lambdas, graph) to one file."""
        st = _L.as_index_save(self._h, gl._h, os.fsencode(path))
        if st:
            _raise(st)

    def query_lambda(self, item, gl: GraphLaplacian) -> float:
        """Extension: lambda_q of `prepare_query_item` (src/lib.rs:154) without the search."""
        q = self._query(item)
        topk = min(int(gl.graph_params["topk"]), self.nitems)
        idx = np.empty(max(topk, 1), dtype=np.int64)
        sc = np.empty(max(topk, 1), dtype=np.float64)
        ln, lq = C.c_int64(0), C.c_double(0.0)
        st = _L.as_search(self._h, gl._h, q.ctypes.data_as(C.c_void_p), q.shape[0], 1.0,
                          idx.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p), C.byref(ln), C.byref(lq))
        if st not in (_lib.AS_OK, _lib.AS_EZEROLAMBDA):
            _raise(st)
        return float(lq.value)

    # out of scope: scorer variants whose semantics are not in the tree
    def search_hybrid(self, item, gl, tau):
        raise NotImplementedError("search_hybrid (src/lib.rs:182-219) is outside the hot path built here")

    def search_energy(self, item, gl, k, w_lambda=None, w_dirichlet=None):
        raise NotImplementedError("search_energy (src/lib.rs:232-262) is outside the hot path built here")


class ArrowSpaceBuilder:
    """src/lib.rs:265-377.  The methods are class methods only so that the reference-named module can carry other
    mode defaults (`_mode`); they are called exactly like the reference's static methods."""

    _mode = NORTH_STAR_MODE

    @classmethod
    def build(cls, graph_params, items):
        """(graph_params: dict|None, items: ndarray[float64, 2-D]) -> (ArrowSpace, GraphLaplacian).
        Argument order as the reference (src/lib.rs:271-275)."""
        if not isinstance(items, np.ndarray) or items.dtype != np.float64 or items.ndim != 2:
            raise TypeError("argument 'items': expected a 2-D numpy.ndarray of dtype float64")
        if items.shape[0] == 0 or items.shape[1] == 0:
            raise ValueError("items must be non-empty 2D array")
        gp, op = _parse_graph_params(graph_params, cls._mode)
        esz = items.itemsize
        rs, cs = items.strides[0] // esz, items.strides[1] // esz
        if items.strides[0] % esz or items.strides[1] % esz or rs < 0 or cs < 0:
            items = np.ascontiguousarray(items)
            rs, cs = items.shape[1], 1
        sp, gr = C.c_void_p(), C.c_void_p()
        st = _L.as_build(items.ctypes.data_as(C.c_void_p), items.shape[0], items.shape[1], rs, cs, C.byref(gp),
                         C.byref(op), C.byref(sp), C.byref(gr))
        if st:
            _raise(st)
        return ArrowSpace._wrap(sp), GraphLaplacian._wrap(gr)

    @classmethod
    def build_from_device(cls, graph_params, data_ptr: int, dtype, n: int, d: int, ld: int | None = None):
        """Extension: items already resident in HBM (row-major fp32 or fp64 at `data_ptr`,
        e.g. `torch_tensor.data_ptr()`); same result as build() on the same values."""
        gp, op = _parse_graph_params(graph_params, cls._mode)
        dt = {"float32": _lib.DTYPE_F32, "float64": _lib.DTYPE_F64}[str(dtype).replace("torch.", "")]
        sp, gr = C.c_void_p(), C.c_void_p()
        st = _L.as_build_dev(C.c_void_p(int(data_ptr)), dt, int(n), int(d), int(ld if ld is not None else d),
                             C.byref(gp), C.byref(op), C.byref(sp), C.byref(gr))
        if st:
            _raise(st)
        return ArrowSpace._wrap(sp), GraphLaplacian._wrap(gr)

    @staticmethod
    def load(path: str):
        """Extension: (ArrowSpace, GraphLaplacian) from a file written by ArrowSpace.save()."""
        op = Opts()
        op.device = int(os.environ.get("ARROWSPACE_DEVICE", "-1"))
        sp, gr = C.c_void_p(), C.c_void_p()
        st = _L.as_index_load(os.fsencode(path), C.byref(op), C.byref(sp), C.byref(gr))
        if st:
            _raise(st)
        return ArrowSpace._wrap(sp), GraphLaplacian._wrap(gr)

    @staticmethod
    def build_energy(items, energy_params=None, graph_params=None):
        raise NotImplementedError("build_energy (src/lib.rs:333-376) is outside the hot path built here")
