"""GPU: the HIP path against the committed golden fixtures (tests/golden) and the
reference-pinned toys, through the arrowspace-compatible surface."""
import json
import os

import numpy as np
import pytest

from conftest import assert_hits_match, clustered

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_readme_example_runs_verbatim():
    """README.md:33-70, literally (module name `arrowspace`, positional call, tau=1.0; no environment variables)."""
    import arrowspace
    from arrowspace import ArrowSpaceBuilder
    assert arrowspace.ArrowSpaceBuilder._mode == {"metric": "cosine", "kernel": "rational", "lambda_mode": "item"}
    t = json.load(open(os.path.join(G, "readme_toy.json")))
    items = np.array(t["items"], dtype=np.float64)
    aspace, gl = ArrowSpaceBuilder.build(t["graph_params"], items)
    hits = aspace.search(np.array(t["query"], dtype=np.float64), gl, t["tau"])
    assert [i for i, _ in hits] == [0, 1, 2]
    np.testing.assert_allclose([s for _, s in hits], [s for _, s in t["expected_hits"]], rtol=1e-12)
    assert aspace.nitems == 3 and aspace.nfeatures == 3 and gl.nnodes == 3 and gl.shape() == (3, 3)
    assert gl.graph_params == t["graph_params"]
    assert aspace.lambdas().shape == (3,)


def test_test0_script_replayed_through_the_reference_named_module(monkeypatch):
    """tests/test_0.py:4-32 replayed through `import arrowspace` with the reference's dict UNMODIFIED and no environment
    variables: the module's default mode is the documented rectified-cosine / rational graph, under which eps = 0.05
    means what the script means.  tau = 1.0: three hits, order [2, 1, 4] (tests/test_0.py:28-32).  The tau < 1 asserts
    (:34-61) are lambda-sensitive and unreachable (profiles/r03_test0_families.md): the search
    runs and returns three hits with item 2 -- lambda_q == lambda_2 -- first."""
    for v in ("ARROWSPACE_METRIC", "ARROWSPACE_KERNEL", "ARROWSPACE_LAMBDA_MODE"):
        monkeypatch.delenv(v, raising=False)
    from arrowspace import ArrowSpaceBuilder, GraphLaplacian  # noqa: F401  (the script's import line)
    t = json.load(open(os.path.join(G, "test0_toy.json")))
    items = np.array(t["items"], dtype=np.float64)
    graph_params = {"eps": 0.05, "k": len(items), "topk": 3, "p": 2.0, "sigma": 0.05}
    assert graph_params == t["graph_params"]
    aspace, gl = ArrowSpaceBuilder.build(graph_params, items)
    query1 = np.array(items[2] * 1.05, dtype=np.float64)
    hits = aspace.search(query1, gl, 1.0)
    assert len(hits) == 3
    assert hits[0][0] == 2
    assert hits[1][0] == 1
    assert hits[2][0] == 4
    for tau in (0.9, 0.6, 0.55):
        hits = aspace.search(query1, gl, tau)
        assert len(hits) == 3 and hits[0][0] == 2


def test_test0_toy_under_the_north_star_default():
    """The same toy through `pyarrowspace_amd` (north_star default: L2 distance): eps = 0.05 leaves no edge and the
    zero-lambda assert fires (src/lib.rs:156-159); with the mode keys in the dict it answers like `arrowspace`."""
    import pyarrowspace_amd as asp
    t = json.load(open(os.path.join(G, "test0_toy.json")))
    items = np.array(t["items"], dtype=np.float64)
    q = np.array(items[2] * 1.05, dtype=np.float64)
    aspace, gl = asp.ArrowSpaceBuilder.build(dict(t["graph_params"], metric="cosine", kernel="rational"), items)
    hits = aspace.search(q, gl, 1.0)
    assert len(hits) == 3 and [i for i, _ in hits] == t["expected_order"]["1.0"]
    aspace2, gl2 = asp.ArrowSpaceBuilder.build(t["graph_params"], items)
    with pytest.raises(asp.PanicException, match="lambdas are zero"):
        aspace2.search(q, gl2, 0.9)


@pytest.mark.parametrize("name", ["synth_64x24_l2", "synth_400x96_cos", "synth_1000x384_l2"])
def test_golden_vectors(name):
    import pyarrowspace_amd as asp
    z = np.load(os.path.join(G, name + ".npz"))
    n, d = int(z["n"]), int(z["d"])
    X = clustered(n, d, nclust=int(z["nclust"]), seed=int(z["seed"]))
    gp = {"eps": float(z["eps"]), "k": int(z["k"]), "topk": int(z["topk"]), "p": 2.0, "sigma": None,
          "metric": str(z["metric"]), "kernel": str(z["kernel"])}
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    np.testing.assert_allclose(aspace.lambdas(), z["lambdas"], rtol=1e-9)
    indptr, indices, values = gl.to_csr()
    rows = np.repeat(np.arange(n), np.diff(indptr))
    off = indices != rows
    assert np.array_equal(indices[off], z["indices"])
    np.testing.assert_allclose(values[off], z["lap"], rtol=1e-9)
    for a, q in enumerate(z["Q"]):
        for b, tau in enumerate(z["taus"]):
            hits = aspace.search(np.ascontiguousarray(q), gl, float(tau))
            want = list(zip(z["hits_idx"][a, b].tolist(), z["hits_score"][a, b].tolist()))
            assert_hits_match(hits, want, rtol=1e-9)   # ties to rounding (tau = 0) may swap


def test_accessors_and_errors():
    """src/lib.rs:100-120,140-146: get_item range error, query length error, strided items."""
    import pyarrowspace_amd as asp
    X = clustered(300, 20, nclust=4, seed=2)
    big = np.zeros((300, 41))
    big[:, ::2][:, :20] = X
    Xs = big[:, ::2][:, :20]                    # non-contiguous view, element strides (41, 2)
    assert not Xs.flags.c_contiguous
    gp = {"eps": 0.9, "k": 5, "topk": 4, "p": 2.0}
    a1, g1 = asp.ArrowSpaceBuilder.build(gp, Xs)
    a2, g2 = asp.ArrowSpaceBuilder.build(gp, np.ascontiguousarray(Xs))
    np.testing.assert_array_equal(a1.lambdas(), a2.lambdas())
    v, lam = a1.get_item(7)
    np.testing.assert_array_equal(v, X[7])
    assert lam == a1.lambdas()[7]
    with pytest.raises(ValueError, match=r"index 300 out of range \[0, 300\)"):
        a1.get_item(300)
    with pytest.raises(ValueError, match="query length 19 must match nfeatures 20"):
        a1.search(np.zeros(19), g1, 0.5)
    with pytest.raises(TypeError):
        a1.search(np.zeros(20, dtype=np.float32), g1, 0.5)
    assert g1.graph_params["sigma"] == 0.45      # eps * 0.5 (src/helpers.rs:68-72)
    far = np.zeros(20)
    far[0] = 50.0
    with pytest.raises(asp.PanicException):
        a1.search(far, g1, 0.5)                  # no neighbour within eps -> lambda_q == 0
    hits = a1.search(np.ascontiguousarray(X[3]), gl=g1, tau=1.0)    # keyword use, tests/test_1_quora_questions.py:108
    assert hits[0][0] == 3 and abs(hits[0][1] - 1.0) < 1e-12 and len(hits) == 4


def test_topk_larger_than_items_and_duplicates():
    import pyarrowspace_amd as asp
    X = np.array([[1.0, 0.0], [1.0, 0.0], [0.0, 1.0], [0.6, 0.8]])
    aspace, gl = asp.ArrowSpaceBuilder.build({"eps": 2.0, "k": 6, "topk": 10, "p": 2.0, "sigma": 1.0}, X)
    hits = aspace.search(np.array([1.0, 0.0]), gl, 1.0)
    assert len(hits) == 4                        # min(topk, nitems)
    assert [i for i, _ in hits][:2] == [0, 1]    # exact tie -> lower index first
