"""ctypes binding of libarrowspace_hip.so (C ABI: include/arrowspace_hip.h).

The product path has no CPU fallback: if the HIP library is missing or cannot be
loaded this module raises ImportError, loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libarrowspace_hip.so")

AS_OK, AS_EINVAL, AS_EZEROLAMBDA, AS_EHIP, AS_EUNSUPPORTED, AS_ENOMEM = range(6)
METRICS = {"l2": 0, "cosine": 1}
KERNELS = {"gaussian": 0, "rational": 1}
LAMBDA_MODES = {"item": 0, "feature": 1}
DTYPE_F32, DTYPE_F64 = 0, 1


class GraphParams(C.Structure):
    _fields_ = [("eps", C.c_double), ("k", C.c_int64), ("topk", C.c_int64), ("p", C.c_double),
                ("sigma", C.c_double), ("has_sigma", C.c_int32), ("_pad", C.c_int32)]


class Opts(C.Structure):
    _fields_ = [("metric", C.c_int32), ("kernel", C.c_int32), ("device", C.c_int32), ("keep_f64", C.c_int32),
                ("force_exact", C.c_int32), ("search_mode", C.c_int32), ("lambda_mode", C.c_int32), ("reserved", C.c_int32)]


class KnnRec(C.Structure):
    _fields_ = [("idx", C.c_int64), ("key", C.c_double), ("dist", C.c_double), ("gy", C.c_double),
                ("deg", C.c_double), ("ny", C.c_double)]


class HitRec(C.Structure):
    _fields_ = [("idx", C.c_int64), ("score", C.c_double)]


# every symbol include/arrowspace_hip.h declares (tests check the .so exports all of them)
SYMBOLS = [
    "as_build", "as_build_dev", "as_space_create_dev", "as_knn_rows", "as_graph_from_knn", "as_feat_gram", "as_feat_graph",
    "as_feat_energy", "as_feat_lambdas", "as_feat_lambdas_global", "as_graph_lambda_mode", "as_knn_list_width", "as_space_nmax", "as_space_norms",
    "as_space_row_offset", "as_knn_block", "as_knn_block_pair", "as_knn_thresholds", "as_knn_merge", "as_knn_fold", "as_knn_block_band", "as_knn_block_exact", "as_record_capacity", "as_graph_from_knn_global", "as_graph_shard_csr", "as_graph_deg_copy", "as_graph_shard_energy",
    "as_graph_energy_copy", "as_graph_shard_lambdas", "as_graph_row_offset", "as_graph_ncols", "as_graph_nitems", "as_search",
    "as_search_batch", "as_search_pool_size", "as_gang_counters", "as_space_knn_pipe", "as_query_scan_int8", "as_last_scan_int8", "as_last_batch_int8", "as_batch_dual_scans", "as_unproven_searches", "as_query_create", "as_query_free", "as_query_scan", "as_query_knn_records",
    "as_query_knn_capacity", "as_query_lambda", "as_query_score", "as_query_hit_records", "as_query_hit_capacity",
    "as_query_finish", "as_query_create_batch", "as_query_slots", "as_query_scan_batch", "as_query_lambda_batch",
    "as_query_score_batch", "as_query_finish_batch", "as_query_set_exact", "as_query_flags", "as_query_stream", "as_query_set_stream", "as_query_bind_records", "as_nitems", "as_nfeatures",
    "as_get_item", "as_lambdas", "as_nnodes", "as_get_graph_params", "as_graph_nnz", "as_graph_csr",
    "as_graph_degrees", "as_graph_tau0", "as_lambdas_dev", "as_build_stats", "as_query_stats", "as_last_search_stats", "as_search_counters", "as_enable_search_stats", "as_set_tuning", "as_index_save", "as_index_load", "as_free_space",
    "as_free_graph", "as_set_debug", "as_last_error", "as_device_count", "as_version",
    "as_comm_available", "as_comm_unique_id", "as_comm_create", "as_comm_free", "as_query_set_comm", "as_query_search_staged", "as_query_x1_bytes", "as_query_x1_usable", "as_query_x1_enabled", "as_query_set_x1", "as_query_x1_begin", "as_query_x1_finish", "as_query_x1_redo", "as_query_set_coarse", "as_query_x1_passes", "as_edges_bucket", "as_ring_i8_stats", "as_ring_i8_set", "as_ring_i8",
]

_lib = None


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64; if this
    library pulled in the system copies first, a later `import torch` would bring up a second runtime that finds
    no GPU ("No HIP GPUs are available").  When torch is installed (not necessarily imported -- importing it
    costs seconds), its bundled runtime is loaded first, so either import order works; without torch the system
    runtime is used as linked."""
    import importlib.util
    import sys

    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    bundled = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(bundled):
        try:
            C.CDLL(bundled, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make` (hipcc --offload-arch=gfx950). "
            "pyarrowspace_amd has no CPU fallback.")
    _preload_torch_hip_runtime()
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise ImportError(f"cannot load {LIB_PATH}: {e}") from e
    vp, i64, i32, f64 = C.c_void_p, C.c_int64, C.c_int32, C.c_double
    pvp = C.POINTER(C.c_void_p)
    pgp, pop = C.POINTER(GraphParams), C.POINTER(Opts)
    sig = {
        "as_build": (i32, [vp, i64, i64, i64, i64, pgp, pop, pvp, pvp]),
        "as_build_dev": (i32, [vp, i32, i64, i64, i64, pgp, pop, pvp, pvp]),
        "as_space_create_dev": (i32, [vp, i32, i64, i64, i64, pop, pvp]),
        "as_knn_rows": (i32, [vp, pgp, i64, i64, vp, vp, vp, vp, vp]),
        "as_graph_from_knn": (i32, [vp, pgp, vp, vp, vp, vp, pvp]),
        "as_knn_list_width": (i32, [i64]),
        "as_space_nmax": (f64, [vp]),
        "as_space_norms": (i32, [vp, vp]),
        "as_space_row_offset": (i64, [vp]),
        "as_knn_block": (i32, [vp, vp, pgp, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp]),
        "as_record_capacity": (i32, [i32]),
        "as_knn_block_exact": (i32, [vp, vp, pgp, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp]),
        "as_knn_block_pair": (i32, [vp, vp, pgp, i64, i64, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "as_knn_thresholds": (i32, [vp, pgp, i64, i64, f64, vp, vp, vp]),
        "as_knn_merge": (i32, [vp, pgp, i64, i64, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(i64)]),
        "as_knn_fold": (i32, [vp, pgp, i64, i64, i32, f64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "as_knn_block_band": (i32, [vp, vp, pgp, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, C.POINTER(i64)]),
        "as_graph_from_knn_global": (i32, [vp, pgp, i64, i64, vp, vp, vp, vp, vp, pvp]),
        "as_graph_shard_csr": (i32, [vp, pgp, i64, i64, vp, vp, vp, vp, i64, vp, vp, vp, vp, pvp]),
        "as_graph_deg_copy": (i32, [vp, vp]),
        "as_graph_shard_energy": (i32, [vp, vp, vp, vp]),
        "as_graph_energy_copy": (i32, [vp, vp]),
        "as_graph_shard_lambdas": (i32, [vp, vp, vp, i64]),
        "as_graph_row_offset": (i64, [vp]),
        "as_graph_ncols": (i64, [vp]),
        "as_graph_nitems": (i64, [vp]),
        "as_feat_gram": (i32, [vp, i64, i64, vp]),
        "as_feat_graph": (i32, [vp, pgp, vp, pvp]),
        "as_feat_energy": (i32, [vp, vp, i64, i64, vp, vp]),
        "as_feat_lambdas": (i32, [vp, vp, vp, vp]),
        "as_feat_lambdas_global": (i32, [vp, vp, vp, vp, i64, i64]),
        "as_graph_lambda_mode": (i32, [vp]),
        "as_search": (i32, [vp, vp, vp, i64, f64, vp, vp, C.POINTER(i64), C.POINTER(f64)]),
        "as_search_batch": (i32, [vp, vp, vp, i64, i64, f64, vp, vp, vp, vp, vp]),
        "as_unproven_searches": (i64, [vp]),
        "as_search_pool_size": (C.c_int32, [vp]),
        "as_space_knn_pipe": (C.c_int32, [vp]),
        "as_query_scan_int8": (C.c_int32, [vp]),
        "as_last_scan_int8": (C.c_int32, [vp]),
        "as_last_batch_int8": (C.c_int32, [vp]),
        "as_batch_dual_scans": (C.c_int64, [vp]),
        "as_query_create": (i32, [vp, vp, pvp]),
        "as_query_free": (None, [vp]),
        "as_query_scan": (i32, [vp, vp, i64, i64, i64]),
        "as_query_knn_records": (vp, [vp]),
        "as_query_knn_capacity": (i64, [vp]),
        "as_query_lambda": (i32, [vp, vp, i64]),
        "as_query_score": (i32, [vp, f64]),
        "as_query_hit_records": (vp, [vp]),
        "as_query_hit_capacity": (i64, [vp]),
        "as_query_finish": (i32, [vp, vp, i64, vp, vp, C.POINTER(i64), C.POINTER(f64)]),
        "as_query_create_batch": (i32, [vp, vp, pvp]),
        "as_query_slots": (i32, [vp]),
        "as_query_scan_batch": (i32, [vp, vp, i32, i64, i64, i64]),
        "as_query_lambda_batch": (i32, [vp, vp, i32]),
        "as_query_score_batch": (i32, [vp, f64]),
        "as_query_finish_batch": (i32, [vp, vp, i32, vp, vp, vp, vp, vp]),
        "as_query_set_exact": (None, [vp, i32]),
        "as_query_flags": (i32, [vp, C.POINTER(i32), C.POINTER(i32)]),
        "as_comm_available": (i32, []),
        "as_comm_unique_id": (i32, [vp]),
        "as_comm_create": (i32, [vp, i32, i32, i32, C.POINTER(vp)]),
        "as_comm_free": (None, [vp]),
        "as_query_set_comm": (i32, [vp, vp]),
        "as_query_x1_bytes": (i64, [vp, i32]),
        "as_query_x1_usable": (i32, [vp, C.c_double]),
        "as_query_x1_enabled": (i32, [vp]),
        "as_query_set_x1": (None, [vp, i32]),
        "as_query_x1_begin": (i32, [vp, vp, i64, i64, i64, C.c_double, vp, i32]),
        "as_query_x1_finish": (i32, [vp, vp, i32, C.c_double, vp, vp, C.POINTER(i64), C.POINTER(C.c_double)]),
        "as_query_x1_redo": (i32, [vp]),
        "as_query_set_coarse": (None, [vp, i32]),
        "as_query_x1_passes": (i64, [vp]),
        "as_ring_i8_stats": (i32, [vp, vp]),
        "as_ring_i8_set": (i32, [vp, C.c_double, C.c_double, i32]),
        "as_ring_i8": (i32, [vp]),
        "as_edges_bucket": (i32, [vp, vp, vp, vp, i64, i64, i64, vp, i32, vp, vp, vp, vp]),
        "as_query_search_staged": (i32, [vp, vp, i64, i64, i64, C.c_double, vp, vp, C.POINTER(i64), C.POINTER(C.c_double)]),
        "as_query_stream": (vp, [vp]),
        "as_query_set_stream": (None, [vp, vp]),
        "as_query_bind_records": (i32, [vp, vp, vp]),
        "as_nitems": (i64, [vp]),
        "as_nfeatures": (i64, [vp]),
        "as_get_item": (i32, [vp, i64, vp, C.POINTER(f64)]),
        "as_lambdas": (i32, [vp, vp]),
        "as_nnodes": (i64, [vp]),
        "as_get_graph_params": (i32, [vp, pgp]),
        "as_graph_nnz": (i64, [vp]),
        "as_graph_csr": (i32, [vp, vp, vp, vp]),
        "as_graph_degrees": (i32, [vp, vp]),
        "as_graph_tau0": (f64, [vp]),
        "as_lambdas_dev": (vp, [vp]),
        "as_build_stats": (i32, [vp, vp, i32]),
        "as_query_stats": (i32, [vp, vp, i32]),
        "as_last_search_stats": (i32, [vp, vp, i32]),
        "as_search_counters": (i32, [vp, vp, i32]),
        "as_gang_counters": (i32, [vp, vp, i32]),
        "as_enable_search_stats": (None, [i32]),
        "as_set_tuning": (i32, [C.c_char_p, i32]),
        "as_index_save": (i32, [vp, vp, C.c_char_p]),
        "as_index_load": (i32, [C.c_char_p, pop, pvp, pvp]),
        "as_free_space": (None, [vp]),
        "as_free_graph": (None, [vp]),
        "as_set_debug": (None, [i32]),
        "as_last_error": (C.c_char_p, []),
        "as_device_count": (i32, []),
        "as_version": (C.c_char_p, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def last_error() -> str:
    return load().as_last_error().decode("utf-8", "replace")
