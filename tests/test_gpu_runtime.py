"""GPU: process-level behaviour -- the library and PyTorch-ROCm share ONE HIP runtime whichever is imported first
(torch wheels bundle their own libamdhip64; a second runtime in the process finds no GPU)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys
sys.path.insert(0, {root!r})
import numpy as np
order = sys.argv[1]
if order == "torch_first":
    import torch
    torch.zeros(1, device="cuda")
import pyarrowspace_amd as asp
X = np.random.default_rng(0).standard_normal((300, 32))
X /= np.linalg.norm(X, axis=1, keepdims=True)
aspace, gl = asp.ArrowSpaceBuilder.build({{"eps": 1.2, "k": 5, "topk": 3, "p": 2.0, "sigma": None}}, X)
assert aspace.search(X[0], gl, 1.0)[0][0] == 0
import torch
assert float(torch.ones(4, device="cuda").sum()) == 4.0
n = sum(1 for l in set(l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l))
assert n == 1, n
print("ok")
"""


@pytest.mark.parametrize("order", ["library_first", "torch_first"])
def test_one_hip_runtime_per_process(order):
    out = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT), order], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stderr[-2000:]


def test_search_is_reentrant_across_host_threads(oracle_lib):
    """as_search from several host threads at once on ONE space.  Every call runs on a pooled workspace (own
    stream, buffers, pinned results), no lock held while it runs: 4 threads x 60 searches (three taus, the exact-item
    query, a far query that must raise the zero-lambda panic in its own thread only) return what the oracle returns and
    what the same searches return single-threaded; the pool has grown to at most 4 workspaces, 1 before the threads ran."""
    import threading

    import numpy as np

    import pyarrowspace_amd as asp
    from conftest import assert_hits_match, calibrate_eps, clustered
    RTOL = 1e-9
    n, d, k, topk = 6000, 96, 9, 7
    X = clustered(n, d, nclust=12, seed=23)
    gp = {"eps": calibrate_eps(X, k, "l2"), "k": k, "topk": topk, "p": 2.0, "sigma": None}
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    ref = oracle_lib.OracleIndex(X, gp)
    rng = np.random.default_rng(5)
    Q = [np.ascontiguousarray(X[rng.integers(0, n)] + 0.03 * rng.standard_normal(d) / np.sqrt(d)) for _ in range(60)]
    Q[7] = np.ascontiguousarray(X[123])
    far = np.zeros(d)
    far[0] = 40.0
    taus = (0.62, 1.0, 0.0)
    single = [aspace.search(q, gl, taus[i % 3]) for i, q in enumerate(Q)]
    assert aspace.search_pool_size == 1
    results = [[None] * len(Q) for _ in range(4)]
    panics, errors = [0] * 4, []

    def worker(t):
        try:
            for i, q in enumerate(Q):
                if (i + t) % 17 == 0:
                    try:
                        aspace.search(far, gl, 0.62)
                    except asp.PanicException:
                        panics[t] += 1
                results[t][i] = aspace.search(q, gl, taus[i % 3])
        except BaseException as e:   # noqa: BLE001
            errors.append(repr(e))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors
    assert all(p >= 3 for p in panics)
    assert 1 <= aspace.search_pool_size <= 4
    for i, q in enumerate(Q):
        want, lq = ref.search(q, taus[i % 3])
        assert_hits_match(single[i], want, ref.scores(q, taus[i % 3], lq), rtol=RTOL)
        for t in range(4):
            assert results[t][i] == single[i], (t, i)


def test_two_graph_handles_on_one_space_from_two_threads():
    """The pool's workspaces carry ONE graph handle's k / topk layout: threads that search the same space against two
    different handles at once must each get their handle's layout (a slot reserved while its workspace is being made
    counts as that handle's: pool_acquire) -- every result equals the single-threaded one, lengths follow the handle's topk."""
    import threading

    import numpy as np

    import pyarrowspace_amd as asp
    from conftest import calibrate_eps, clustered
    n, d = 5000, 64
    X = clustered(n, d, nclust=10, seed=31)
    eps = calibrate_eps(X, 8, "l2")
    gp1 = {"eps": eps, "k": 8, "topk": 5, "p": 2.0, "sigma": None}
    gp2 = {"eps": eps, "k": 12, "topk": 40, "p": 2.0, "sigma": None}
    aspace, gl1 = asp.ArrowSpaceBuilder.build(gp1, X)
    _other, gl2 = asp.ArrowSpaceBuilder.build(gp2, X)
    rng = np.random.default_rng(9)
    Q = [np.ascontiguousarray(X[rng.integers(0, n)] + 0.03 * rng.standard_normal(d) / np.sqrt(d)) for _ in range(40)]
    want1 = [aspace.search(q, gl1, 0.62) for q in Q]
    want2 = [aspace.search(q, gl2, 0.62) for q in Q]
    assert all(len(h) == 5 for h in want1) and all(len(h) == 40 for h in want2)
    errors = []

    def worker(gl, want):
        try:
            for _ in range(3):
                for q, w in zip(Q, want):
                    got = aspace.search(q, gl, 0.62)
                    if got != w:
                        errors.append((len(got), len(w)))
        except BaseException as e:   # noqa: BLE001
            errors.append(repr(e))

    ths = [threading.Thread(target=worker, args=(gl1, want1)), threading.Thread(target=worker, args=(gl2, want2)),
           threading.Thread(target=worker, args=(gl1, want1)), threading.Thread(target=worker, args=(gl2, want2))]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors[:4]


def test_concurrent_callers_share_scans_and_get_the_serial_answers():
    """Gang scans: as_search callers that arrive together are served by ONE pass over the items (up to four queries per launch,
    scan_tile_gang_kernel).  Native threads against the C ABI (tools/probe/thread_driver.cpp; no interpreter lock between their
    calls) on a 300 000 x 128 index: every call's best hit is the serial one, multi-member scans did form, and Python threads
    on the same space still get the serial answers element for element."""
    import threading

    import numpy as np

    import pyarrowspace_amd as asp
    from conftest import calibrate_eps, clustered
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import thread_bench
    asp.enable_search_stats(False)     # (process-global; a scan that is timed per launch is not shared)
    n, d = 300000, 256
    X = clustered(n, d, nclust=1024, seed=5)
    gp = {"eps": calibrate_eps(X, 10, "l2"), "k": 10, "topk": 8, "p": 2.0, "sigma": None}
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    rng = np.random.default_rng(11)
    Q = np.ascontiguousarray(X[rng.integers(0, n, 96)] + 0.02 * rng.standard_normal((96, d)) / np.sqrt(d))
    want = [aspace.search(q, gl, 0.62) for q in Q]
    assert aspace.last_scan_operand == "int8-high"
    first = np.array([w[0][0] for w in want], dtype=np.int64)
    shared = 0
    for nthr in (2, 4, 3, 4):
        rate, errs, gangs = thread_bench.native_rate(aspace, gl, Q, 0.62, nthr, 150, first)
        assert errs == 0 and rate > 0
        shared += sum(gangs[1:])
    # scans with two or more members did form (how many is a matter of timing: a search whose candidates overflowed sends its
    # workspace's next 63 through the coarse chain, which is not shared)
    assert shared > 0, (aspace.gang_counters(), aspace.gang_skips(), aspace.search_counters())
    bad = []

    def worker(t):
        for i in range(120):
            j = (t * 31 + i) % len(Q)
            if aspace.search(Q[j], gl, 0.62) != want[j]:
                bad.append((t, i))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not bad, bad[:5]
    c = aspace.search_counters()
    assert c["searches_with_rerun"] <= 0.02 * c["searches"]


def test_batched_calls_beside_concurrent_single_searches():
    """One thread in `search_batch` (pairs of passes sharing a scan, two pairs of workspaces alternating) while three threads run
    single searches on the same space (pooled workspaces, gang scans): every result equals the serial one.  The batched
    workspaces are the space's own (one batched call at a time: `bmu`), the single searches take workspaces from the pool --
    nothing is shared between them but the items and the graph."""
    import threading

    import numpy as np

    import pyarrowspace_amd as asp
    from conftest import calibrate_eps, clustered
    n, d = 120000, 256
    X = clustered(n, d, nclust=400, seed=17)
    gp = {"eps": calibrate_eps(X, 10, "l2"), "k": 10, "topk": 8, "p": 2.0, "sigma": None}
    aspace, gl = asp.ArrowSpaceBuilder.build(gp, X)
    rng = np.random.default_rng(23)
    Q = np.ascontiguousarray(X[rng.integers(0, n, 160)] + 0.02 * rng.standard_normal((160, d)) / np.sqrt(d))
    want = [aspace.search(q, gl, 0.62) for q in Q]
    assert aspace.search_batch(Q, gl, 0.62) == want
    bad, errors = [], []

    def batcher():
        try:
            for rep in range(12):
                lo = (rep * 7) % 40
                got = aspace.search_batch(Q[lo : lo + 120], gl, 0.62)
                if got != want[lo : lo + 120]:
                    bad.append(("batch", rep))
        except BaseException as e:   # noqa: BLE001
            errors.append(repr(e))

    def single(t):
        try:
            for i in range(150):
                j = (t * 37 + i) % len(Q)
                if aspace.search(Q[j], gl, 0.62) != want[j]:
                    bad.append(("single", t, i))
        except BaseException as e:   # noqa: BLE001
            errors.append(repr(e))

    ths = [threading.Thread(target=batcher)] + [threading.Thread(target=single, args=(t,)) for t in range(3)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not errors, errors[:3]
    assert not bad, bad[:5]
    assert aspace.batch_dual_scans > 0
