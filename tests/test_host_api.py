"""CPU: host-side mirror of the reference interface (names, argument checks, errors) and
the C ABI: libarrowspace_hip.so loads and exports every symbol include/arrowspace_hip.h
declares.  No compute call is made without a GPU; the product has no CPU fallback."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def asp():
    import __graft_entry__ as g
    g.build()
    import pyarrowspace_amd
    return pyarrowspace_amd


def test_library_exports_every_declared_symbol(asp):
    hdr = open(os.path.join(ROOT, "include", "arrowspace_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(as_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 35
    lib = ctypes.CDLL(asp._lib.LIB_PATH)
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(asp._lib.SYMBOLS) == declared, set(asp._lib.SYMBOLS) ^ set(declared)


def test_module_surface_matches_reference(asp):
    """src/lib.rs:379-386: three classes + set_debug under the module name `arrowspace`."""
    import arrowspace
    for name in ("ArrowSpaceBuilder", "ArrowSpace", "GraphLaplacian", "set_debug"):
        assert hasattr(arrowspace, name)
    # the reference-named module carries the documented graph as its default mode, the package the north_star's
    assert issubclass(arrowspace.ArrowSpaceBuilder, asp.ArrowSpaceBuilder)
    assert arrowspace.ArrowSpaceBuilder._mode == asp.REFERENCE_MODE == {"metric": "cosine", "kernel": "rational", "lambda_mode": "item"}
    assert asp.ArrowSpaceBuilder._mode == asp.NORTH_STAR_MODE == {"metric": "l2", "kernel": "gaussian", "lambda_mode": "item"}
    assert arrowspace.ArrowSpace is asp.ArrowSpace and arrowspace.GraphLaplacian is asp.GraphLaplacian
    arrowspace.set_debug(True)
    arrowspace.set_debug(False)


def test_direct_construction_raises_valueerror(asp):
    """src/lib.rs:33-38,71-76."""
    with pytest.raises(ValueError, match="cannot be constructed directly"):
        asp.ArrowSpace()
    with pytest.raises(ValueError, match="cannot be constructed directly"):
        asp.GraphLaplacian()


def test_build_argument_checks(asp):
    gp = {"eps": 1.0, "k": 3, "topk": 2, "p": 2.0}
    X = np.zeros((4, 3))
    with pytest.raises(TypeError):
        asp.ArrowSpaceBuilder.build(gp, X.astype(np.float32))      # PyReadonlyArray2<f64>
    with pytest.raises(TypeError):
        asp.ArrowSpaceBuilder.build(gp, np.zeros(3))
    with pytest.raises(ValueError, match="non-empty"):
        asp.ArrowSpaceBuilder.build(gp, np.zeros((0, 3)))            # src/helpers.rs:27-29
    for key in ("eps", "k", "topk", "p"):
        bad = dict(gp)
        del bad[key]
        with pytest.raises(ValueError, match=key):                   # src/helpers.rs:52-67
            asp.ArrowSpaceBuilder.build(bad, np.ones((4, 3)))
    with pytest.raises(ValueError, match="metric"):
        asp.ArrowSpaceBuilder.build(dict(gp, metric="manhattan"), np.ones((4, 3)))
    with pytest.raises(NotImplementedError):
        asp.ArrowSpaceBuilder.build_energy(np.ones((4, 3)))


def test_sigma_default_is_half_eps(asp):
    gp, _ = asp._parse_graph_params({"eps": 0.5, "k": 3, "topk": 2, "p": 2.0, "sigma": None})
    assert gp.has_sigma == 0          # resolved to eps*0.5 inside the library (src/helpers.rs:68-72)
    gp, _ = asp._parse_graph_params({"eps": 0.5, "k": 3, "topk": 2, "p": 2.0, "sigma": 0.1})
    assert gp.has_sigma == 1 and gp.sigma == 0.1


def test_no_cpu_fallback_without_a_gpu(asp):
    """On a box without a device the build fails loudly instead of computing on the CPU."""
    if asp._lib.load().as_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError):
        asp.ArrowSpaceBuilder.build({"eps": 1.0, "k": 2, "topk": 2, "p": 2.0}, np.eye(3))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pyarrowspace_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.lower() or f == "__init__.py" and "oracle" not in txt, (dirpath, f)


def test_sharded_index_refuses_record_counts_beyond_the_merge_capacity(asp):
    """world * k neighbour records and world * (topk + 1) hit records are merged by one workgroup per query (1 024 and
    8 208): a combination beyond that is refused before anything is built, not at the first search."""
    from pyarrowspace_amd.dist import check_world_limits
    gp = {"eps": 1.0, "k": 120, "topk": 15, "p": 2.0, "sigma": None}
    check_world_limits(gp, 8)                                   # 960 records
    with pytest.raises(ValueError, match="neighbour records"):
        check_world_limits(gp, 9)                               # 1 080
    check_world_limits(dict(gp, lambda_mode="feature"), 10)    # no neighbour records in feature mode
    check_world_limits(dict(gp, k=56), 18)                     # 1 008
    check_world_limits(dict(gp, k=25, topk=1024), 8)           # 8 200 hit records
    with pytest.raises(ValueError, match="hit records"):
        check_world_limits(dict(gp, k=25, topk=1024), 9)


def test_slice_message_sections_tile_the_flat_buffer():
    """The symmetric ring's slice message is ONE flat fp64 buffer whose sections (keys, distances, y.y, ids, counts,
    bounds) the library writes in place: contiguous views that do not overlap, cover the buffer, and are 8-byte aligned,
    for odd row counts and all list widths."""
    import torch

    from pyarrowspace_amd.dist import HipEngine
    for M in (32, 64, 128):
        for n in (1, 2, 3, 7, 128, 1001):
            e = HipEngine.__new__(HipEngine)
            e.torch, e.M = torch, M
            shape = e.slice_shape(n)
            P = torch.zeros(shape, dtype=torch.float64)
            sec = e._slice_sections(P, n)
            assert [tuple(s.shape) for s in sec] == [(n, M), (n, M), (n, M), (n, M), (n,), (n,)]
            assert [s.dtype for s in sec] == [torch.float64] * 3 + [torch.int32, torch.int32, torch.float32]
            assert all(s.is_contiguous() and s.data_ptr() % 8 == 0 for s in sec)
            spans = sorted((s.data_ptr() - P.data_ptr(), s.data_ptr() - P.data_ptr() + s.numel() * s.element_size()) for s in sec)
            assert spans[0][0] == 0 and all(a[1] <= b[0] for a, b in zip(spans, spans[1:])) and spans[-1][1] <= P.numel() * 8
            for v, s in enumerate(sec):      # writing one section touches no other
                s.fill_(v + 1)
            assert all(bool((s == v + 1).all()) for v, s in enumerate(sec))
            assert e.slice_shape(0) == e.slice_shape(1)


def test_product_library_has_no_wrong_answer_switches():
    """Measurement switches that make kernels return wrong results (ARROWSPACE_SC_DBG: the fused tail's cost breakdown;
    ARROWSPACE_GEMM_VARIANT=16 and ARROWSPACE_K2_DIAG: no-MFMA / no-DMA timing skeletons; ARROWSPACE_K2_ZERO: all-zero
    operands) are compiled into `make ABLATION=1` builds only: the product library does not even know their names, so a
    stray environment variable cannot turn a serving process's answers into garbage."""
    import os
    lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pyarrowspace_amd", "libarrowspace_hip.so")
    blob = open(lib, "rb").read()
    for name in (b"ARROWSPACE_SC_DBG", b"ARROWSPACE_K2_DIAG", b"ARROWSPACE_K2_ZERO", b"ARROWSPACE_KNN_VARIANT_ABLATION"):
        assert name not in blob, name.decode() + " is compiled into the product library (was it built with ABLATION=1?)"
    assert b"ARROWSPACE_K2_FP32" in blob     # (the names of the switches that DO exist are in there: the scan is meaningful)
    # the batched scan's no-MFMA skeleton is an extra kernel instantiation of ablation builds
    assert b"scan_gemm_kernelILi4ELi1ELi2E" not in blob
