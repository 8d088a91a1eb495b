"""GPU: the symmetric k-NN build (whole-index builds compute only the column tiles at or above each row block; a key
above the diagonal reaches the column item's row through its transposed buffer, gated by per-item thresholds from a
pass over every 8th column tile) against the full pass (ARROWSPACE_NO_SYM=1): bit-identical graphs and lambdas, fewer
MFMA flops issued; a transposed buffer that overflows sends its row to the band pass; an eps that admits every pair
works the same way."""
import os

import numpy as np
import pytest

from conftest import calibrate_eps, clustered

pytestmark = pytest.mark.gpu


def _build(X, gp, sym, chunks=None):
    """chunks: the symmetric main pass in that many column chunks (ARROWSPACE_SYM_CHUNKS), each with the transposed buffers
    of its own items only -- what a shard whose buffers do not fit the free memory gets."""
    from pyarrowspace_amd import ArrowSpaceBuilder
    old = os.environ.pop("ARROWSPACE_NO_SYM", None)
    oldc = os.environ.pop("ARROWSPACE_SYM_CHUNKS", None)
    if not sym:
        os.environ["ARROWSPACE_NO_SYM"] = "1"
    if chunks is not None:
        os.environ["ARROWSPACE_SYM_CHUNKS"] = str(chunks)
    try:
        aspace, gl = ArrowSpaceBuilder.build(gp, X)
    finally:
        os.environ.pop("ARROWSPACE_NO_SYM", None)
        os.environ.pop("ARROWSPACE_SYM_CHUNKS", None)
        if old is not None:
            os.environ["ARROWSPACE_NO_SYM"] = old
        if oldc is not None:
            os.environ["ARROWSPACE_SYM_CHUNKS"] = oldc
    return aspace, gl, gl.build_stats()


def _same_index(a, b):
    (sa, ga, _), (sb, gb, _) = a, b
    np.testing.assert_array_equal(np.asarray(sa.lambdas()), np.asarray(sb.lambdas()))
    for u, v in zip(ga.to_csr(), gb.to_csr()):
        np.testing.assert_array_equal(u, v)


@pytest.mark.parametrize("metric,kernel", [("l2", "gaussian"), ("cosine", "rational")])
@pytest.mark.parametrize("n,d,k,ratio", [(5000, 96, 10, 0.85), (3000, 768, 25, 0.95), (20000, 64, 10, 0.7), (1100, 40, 6, None)])
def test_symmetric_pass_equals_the_full_pass_bitwise(metric, kernel, n, d, k, ratio):
    X = clustered(n, d, nclust=8, seed=4)
    gp = {"eps": calibrate_eps(X, k, metric), "k": k, "topk": 5, "p": 2.0, "sigma": None, "metric": metric, "kernel": kernel}
    sym, full = _build(X, gp, True), _build(X, gp, False)
    _same_index(sym, full)
    if ratio is None:   # fewer than 16 column tiles: the full pass either way
        assert sym[2]["mfma_flops"] == full[2]["mfma_flops"]
    else:               # the triangle plus the threshold pass, not the square
        assert sym[2]["mfma_flops"] < ratio * full[2]["mfma_flops"]
    assert sym[2]["fallback_rows"] == 0


def test_overflowing_transposed_buffers_go_to_the_band_pass():
    """900 exact copies of one item at the end of the index: every copy is at distance 0 from the 899 others, the
    thresholds cannot separate them, and the copies in the last row blocks receive more transposed entries than a
    buffer holds (16 M = 512): those rows are flagged and settled exactly by the band pass."""
    n, d, k = 6000, 64, 12
    X = clustered(n, d, nclust=12, seed=8, normalise=False)
    X[5100:6000] = X[5100]
    eps = calibrate_eps(X[:5000], k)
    gp = {"eps": eps, "k": k, "topk": 5, "p": 2.0, "sigma": None}
    sym, full = _build(X, gp, True), _build(X, gp, False)
    _same_index(sym, full)
    assert sym[2]["mfma_flops"] < full[2]["mfma_flops"]
    assert sym[2]["band_rows"] >= 100 and sym[2]["fallback_rows"] == 0


@pytest.mark.parametrize("metric", ["l2", "cosine"])
def test_an_eps_that_admits_every_pair(metric):
    """eps = 10 (tests/test_3_beir.py's literal): only the thresholds keep the transposed buffers small."""
    n, d, k = 9000, 64, 8
    X = clustered(n, d, nclust=6, seed=2)
    gp = {"eps": 10.0, "k": k, "topk": 5, "p": 2.0, "sigma": None, "metric": metric}
    sym, full = _build(X, gp, True), _build(X, gp, False)
    _same_index(sym, full)
    assert sym[2]["mfma_flops"] < 0.8 * full[2]["mfma_flops"]


@pytest.mark.parametrize("scale", [0.7, 1.0, 2.0])
def test_symmetric_pass_at_100k(scale):
    """100 000 x 128, k = 25 (M = 64, every 64th tile sampled), eps tighter and looser than calibrated: bit-identical
    graphs, well under the full pass's flops."""
    import torch

    from conftest import gpu_clustered
    n, d, k = 100_000, 128, 25
    X = gpu_clustered(n, d, 77, nclust=256).double().cpu().numpy()
    eps = calibrate_eps(X[:20000], k) * scale
    gp = {"eps": eps, "k": k, "topk": 5, "p": 2.0, "sigma": None}
    sym, full = _build(X, gp, True), _build(X, gp, False)
    _same_index(sym, full)
    assert sym[2]["mfma_flops"] < 0.62 * full[2]["mfma_flops"]
    assert sym[2]["fallback_rows"] == 0
    torch.cuda.empty_cache()


@pytest.mark.parametrize("metric", ["l2", "cosine"])
def test_symmetric_pass_with_zero_rows_duplicates_and_unrepresentable_items(metric):
    """All-zero rows (under cosine they tie with everything at distance 1), groups of exact duplicates and fp64 items that
    do not round-trip through fp32 (the exact re-evaluation then reads the fp64 copy): still bit-identical to the full
    pass."""
    n, d, k = 5200, 48, 9
    X = clustered(n, d, nclust=10, seed=12) * (1.0 + 1e-9 * np.arange(n)[:, None])     # not fp32-representable
    rng = np.random.default_rng(4)
    X[rng.choice(n, 40, replace=False)] = 0.0
    for g in range(5):
        rows = rng.choice(n, 30, replace=False)
        X[rows] = X[rows[0]]
    gp = {"eps": calibrate_eps(clustered(n, d, nclust=10, seed=12), k, metric), "k": k, "topk": 5, "p": 2.0, "sigma": None,
          "metric": metric}
    sym, full = _build(X, gp, True), _build(X, gp, False)
    _same_index(sym, full)
    assert sym[2]["mfma_flops"] < full[2]["mfma_flops"] and sym[2]["fallback_rows"] == full[2]["fallback_rows"]


@pytest.mark.parametrize("metric,kernel", [("l2", "gaussian"), ("cosine", "rational")])
@pytest.mark.parametrize("n,d,k", [(5000, 96, 10), (20000, 64, 10), (41000, 48, 25)])
def test_column_chunked_symmetric_pass_is_bitwise_the_same(metric, kernel, n, d, k):
    """The main pass in 2, 3 and 7 column chunks (a shard of 8M rows takes 2: DESIGN.md section 6): the units are cut at
    multiples of the piece length instead of at each row block's diagonal, a chunk's transposed buffers hold its own
    items only -- same graph and lambdas bit for bit, same triangle of tiles."""
    X = clustered(n, d, nclust=8, seed=4)
    gp = {"eps": calibrate_eps(X, k, metric), "k": k, "topk": 5, "p": 2.0, "sigma": None, "metric": metric, "kernel": kernel}
    one = _build(X, gp, True)
    for chunks in (2, 3, 7):
        got = _build(X, gp, True, chunks)
        _same_index(got, one)
        # (the triangle's tiles are the same; which rows a band pass settles afterwards depends on how the lists are segmented)
        assert abs(got[2]["mfma_flops"] - one[2]["mfma_flops"]) < 0.05 * one[2]["mfma_flops"] and got[2]["fallback_rows"] == 0


def test_column_chunks_with_overflowing_buffers_and_a_loose_eps():
    """The two hard cases of this file again, chunked: 900 copies of one item (buffers overflow -> band pass) and eps = 10."""
    n, d, k = 6000, 64, 12
    X = clustered(n, d, nclust=12, seed=8, normalise=False)
    X[5100:6000] = X[5100]
    gp = {"eps": calibrate_eps(X[:5000], k), "k": k, "topk": 5, "p": 2.0, "sigma": None}
    one, got = _build(X, gp, True), _build(X, gp, True, 3)
    _same_index(got, one)
    assert got[2]["band_rows"] >= 100 and got[2]["fallback_rows"] == 0
    X = clustered(9000, 64, nclust=6, seed=2)
    for metric in ("l2", "cosine"):
        gp = {"eps": 10.0, "k": 8, "topk": 5, "p": 2.0, "sigma": None, "metric": metric}
        _same_index(_build(X, gp, True, 4), _build(X, gp, True))


def _build_pipe(X, gp, fp32, **env):
    """One build on the bf16 head + tail kernel (default) or, ARROWSPACE_K2_FP32=1, on the fp32 matrix pipe."""
    keys = dict(env, ARROWSPACE_K2_FP32="1" if fp32 else None)
    old = {k: os.environ.pop(k, None) for k in keys}
    for k, v in keys.items():
        if v is not None:
            os.environ[k] = str(v)
    try:
        return _build(X, gp, True)
    finally:
        for k in keys:
            os.environ.pop(k, None)
            if old[k] is not None:
                os.environ[k] = old[k]


@pytest.mark.parametrize("metric,kernel", [("l2", "gaussian"), ("cosine", "rational")])
@pytest.mark.parametrize("n,d,k", [(6000, 96, 10), (3000, 768, 25), (20000, 64, 10), (1100, 40, 6), (9000, 1000, 57)])
def test_bf16_build_kernel_gives_the_graphs_of_the_fp32_pipe_bitwise(metric, kernel, n, d, k):
    """The k-NN block of the build on the bf16 matrix pipe (as_k2bf.hip: every operand as head + tail, three products) is a
    PREFILTER like the fp32 kernel it replaces: the refinement evaluates what it keeps exactly and proves what it dropped
    against the wider error term (err_coef) -- so CSR, Laplacian values and lambdas are bit for bit those of the fp32 pipe
    (ARROWSPACE_K2_FP32=1), symmetric pass, threshold pass and band pass included; also in gang order (per-XCD unit lists)."""
    X = clustered(n, d, nclust=8, seed=4)
    gp = {"eps": calibrate_eps(X, k, metric), "k": k, "topk": 5, "p": 2.0, "sigma": None, "metric": metric, "kernel": kernel}
    bf, f32 = _build_pipe(X, gp, False), _build_pipe(X, gp, True)
    _same_index(bf, f32)
    assert bf[2]["fallback_rows"] == 0 and bf[2]["mfma_flops"] > 0
    if n >= 6000:
        _same_index(_build_pipe(X, gp, False, ARROWSPACE_K2_GANG_GC=4), f32)


def test_bf16_build_kernel_with_duplicates_zero_rows_and_scaled_items():
    """What strains the wider error term: groups of exact duplicates (ties at the k-th distance: band pass), zero rows, items
    scaled by 1e3 and 1e-3 in one index (norms six decades apart), values that do not round-trip through fp32."""
    rng = np.random.default_rng(9)
    X = clustered(7000, 128, nclust=6, seed=11, normalise=False)
    X[100:140] = X[100]            # 40 copies
    X[2000:2003] = 0.0
    X[3000:3500] *= 1.0e3
    X[4000:4500] *= 1.0e-3
    X[5000:5200] += 1.0e-9 * rng.standard_normal((200, 128))
    for metric in ("l2", "cosine"):
        gp = {"eps": calibrate_eps(X, 12, metric), "k": 12, "topk": 5, "p": 2.0, "sigma": None, "metric": metric}
        bf, f32 = _build_pipe(X, gp, False), _build_pipe(X, gp, True)
        _same_index(bf, f32)
