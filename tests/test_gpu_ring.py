"""GPU: k-NN over visiting column blocks (the multi-GPU ring's per-rank work) on one device -- the items cut into
uneven blocks, every block a temporary space, as_knn_block per block, as_knn_merge, and the collect-mode second round
for rows the merge cannot prove exact -- against as_knn_rows on one space holding everything (bit-identical lists)
and against the oracle."""
import numpy as np
import pytest

from conftest import calibrate_eps, clustered

pytestmark = pytest.mark.gpu


def _ring_lists(X, gp, cuts, own):
    """Lists of the rows of block `own` against all blocks, through HipEngine's ring methods."""
    import torch

    from pyarrowspace_amd.dist import HipEngine
    e = HipEngine(gp)
    lo, hi = cuts[own], cuts[own + 1]
    e.create_space(torch.from_numpy(X[lo:hi].copy()).cuda())
    e.ring_begin(len(cuts) - 1)
    blocks = [torch.from_numpy(X[cuts[b]:cuts[b + 1]].copy()).cuda() for b in range(len(cuts) - 1)]
    nmax = []

    def round_(fn):
        for b, Xb in enumerate(blocks):
            h = e.own_block() if b == own else e.open_block(Xb)
            if len(nmax) < len(blocks):
                nmax.append(e.block_nmax(h))
            fn(h, b, lo, cuts[b])
            if b != own:
                e.close_block(h)

    round_(e.knn_block)
    nflag = e.knn_merge(nmax)
    over = 0
    if nflag:
        res = []
        round_(lambda h, b, rg, cg: res.append(e.knn_block_band(h, b, rg, cg)))
        over = sum(res)
        e.knn_merge(nmax)
        if e.overflowed_rows():          # third round: exact evaluation of every pair for the rows whose band overflowed
            round_(e.knn_block_exact)
            e.knn_merge(nmax, final=True)
    idx, dist, gy, cnt = [t.cpu().numpy() for t in e.lists()]
    key = e.l_key[: hi - lo].cpu().numpy()
    e.close()
    return idx, key, dist, gy, cnt, nflag, over


def _single_lists(X, gp, lo, hi):
    import torch

    from pyarrowspace_amd.dist import HipEngine
    e = HipEngine(gp)
    e.create_space(torch.from_numpy(X).cuda())
    idx, dist, gy, cnt = [t.cpu().numpy() for t in e.knn_rows(lo, hi)]
    e.close()
    return idx, dist, gy, cnt


@pytest.mark.parametrize("metric", ["l2", "cosine"])
@pytest.mark.parametrize("n,d,k,cuts", [(3000, 96, 10, [0, 700, 1900, 3000]), (5000, 768, 25, [0, 2500, 5000]), (900, 40, 6, [0, 100, 101, 600, 900])])
def test_blockwise_lists_equal_the_single_space_lists(oracle_lib, metric, n, d, k, cuts):
    X = clustered(n, d, nclust=max(4, n // 200), seed=n + k)
    gp = {"eps": calibrate_eps(X, k, metric), "k": k, "topk": 5, "p": 2.0, "sigma": None, "metric": metric}
    ref = oracle_lib.OracleIndex(X, gp)
    for own in range(len(cuts) - 1):
        lo, hi = cuts[own], cuts[own + 1]
        idx, key, dist, gy, cnt, nflag, over = _ring_lists(X, gp, cuts, own)
        sidx, sdist, sgy, scnt = _single_lists(X, gp, lo, hi)
        assert over == 0
        np.testing.assert_array_equal(cnt, scnt)
        np.testing.assert_array_equal(idx, sidx)             # global ids, same order
        np.testing.assert_array_equal(dist, sdist)           # exact keys are evaluated the same way: same bits
        np.testing.assert_array_equal(gy, sgy)
        np.testing.assert_array_equal(cnt, ref.knn_cnt[lo:hi])
        np.testing.assert_array_equal(idx, ref.knn_idx[lo:hi])


def test_blockwise_second_round_settles_duplicate_groups(oracle_lib):
    """Groups of 150 identical rows spread over the blocks (more per block than a candidate list is wide): the merge
    flags them (ties at the k-th distance), the second round collects each block's band, and the lists equal the
    oracle's (ties by global index)."""
    rng = np.random.default_rng(3)
    n, d, k = 2400, 64, 8
    X = clustered(n, d, nclust=12, seed=17)
    for g in range(6):
        rows = rng.choice(n, 150, replace=False)
        X[rows] = X[rows[0]]
    gp = {"eps": calibrate_eps(clustered(n, d, nclust=12, seed=17), k), "k": k, "topk": 5, "p": 2.0, "sigma": None}
    ref = oracle_lib.OracleIndex(X, gp)
    cuts = [0, 800, 1700, 2400]
    flagged = 0
    for own in range(3):
        lo, hi = cuts[own], cuts[own + 1]
        idx, key, dist, gy, cnt, nflag, over = _ring_lists(X, gp, cuts, own)
        flagged += nflag
        assert over == 0
        np.testing.assert_array_equal(cnt, ref.knn_cnt[lo:hi])
        np.testing.assert_array_equal(idx, ref.knn_idx[lo:hi])
    assert flagged > 0          # the second round did run


def _symmetric_ring_lists(X, gp, cuts, dup_round=True, i8=False):
    """The symmetric ring's first round with every rank in this process: each unordered pair of blocks goes through
    as_knn_block_pair ONCE (the even world's opposite pair split as ShardedIndex._ring_round_symmetric splits it), the
    visiting rows' slice is handed to their engine and folded there; then merge, and the second round for flagged rows
    exactly as the ranks would run it (every block visits, collect mode)."""
    import torch

    from pyarrowspace_amd.dist import HipEngine
    G = len(cuts) - 1
    counts = [cuts[b + 1] - cuts[b] for b in range(G)]
    blocks = [torch.from_numpy(X[cuts[b]:cuts[b + 1]].copy()).cuda() for b in range(G)]
    eng = []
    for b in range(G):
        e = HipEngine(gp)
        e.create_space(blocks[b])
        e.ring_begin(G)
        eng.append(e)
    nmax = [e.block_nmax(e.own_block()) for e in eng]
    if i8:   # the ranks' agreement on the int8 images (ShardedIndex._ring_knn: one all-gather of three numbers)
        st = np.array([e.ring_i8_stats() for e in eng])
        assert all(e.ring_i8_set(st[:, 0].max(), st[:, 1].max(), st[:, 2].max() == 0.0) for e in eng)
    for r, e in enumerate(eng):
        e.knn_block(e.own_block(), r, cuts[r], cuts[r])
    U = [e.knn_thresholds(max(nmax)) for e in eng]
    half = G // 2
    for s in range(1, half + 1):
        out = {}
        for r, e in enumerate(eng):
            src, dst = (r - s) % G, (r + s) % G
            row0, row1, ct0, ct1 = 0, counts[r], -1, -1
            if 2 * s == G:
                q = max(r, src)
                tq = (counts[q] + 255) // 256 * 256 // 128
                if r == q:
                    row0 = min(counts[q], (tq // 2) * 128)
                else:
                    ct0, ct1 = 0, tq // 2
            h = e.open_block(blocks[src])
            out[(r, src)] = e.knn_block_pair(h, row0, row1, ct0, ct1, cuts[r], cuts[src], U[src], counts[src], row_thr=U[r])
            e.close_block(h)
        for (r, src), P in out.items():
            eng[src].fold_slice(P, nmax[r])          # the slice of src's rows that rank r computed goes home
    res, flagged = [], 0
    for r, e in enumerate(eng):
        nflag = e.knn_merge(nmax)
        flagged += nflag
        if nflag and dup_round:
            for b in range(G):
                h = e.own_block() if b == r else e.open_block(blocks[b])
                e.knn_block_band(h, b, cuts[r], cuts[b])
                if b != r:
                    e.close_block(h)
            e.knn_merge(nmax)
            if e.overflowed_rows():
                for b in range(G):
                    h = e.own_block() if b == r else e.open_block(blocks[b])
                    e.knn_block_exact(h, b, cuts[r], cuts[b])
                    if b != r:
                        e.close_block(h)
                e.knn_merge(nmax, final=True)
        res.append([t.cpu().numpy() for t in e.lists()])
    for e in eng:
        e.close()
    return res, flagged


@pytest.mark.parametrize("i8", [False, True], ids=["bf16-ring", "int8-ring"])
@pytest.mark.parametrize("metric", ["l2", "cosine"])
@pytest.mark.parametrize("n,d,k,cuts", [(3000, 96, 10, [0, 700, 1900, 3000]), (5000, 768, 25, [0, 2500, 5000]),
                                        (2600, 40, 6, [0, 500, 501, 1700, 2600]), (4100, 64, 12, [0, 300, 1400, 2000, 3300, 4100])])
def test_symmetric_ring_lists_equal_the_single_space_lists(metric, n, d, k, cuts, i8):
    """3, 2, 4 and 5 blocks (uneven, one of a single row): whole pairs, the split opposite pair of the even worlds; the block
    passes on the bf16 head + tail images and on the shards' int8 images (the ring-wide coefficient of as_ring_i8_set)."""
    X = clustered(n, d, nclust=max(4, n // 200), seed=n + k)
    gp = {"eps": calibrate_eps(X, k, metric), "k": k, "topk": 5, "p": 2.0, "sigma": None, "metric": metric}
    res, _ = _symmetric_ring_lists(X, gp, cuts, i8=i8)
    for b in range(len(cuts) - 1):
        lo, hi = cuts[b], cuts[b + 1]
        idx, dist, gy, cnt = res[b]
        sidx, sdist, sgy, scnt = _single_lists(X, gp, lo, hi)
        np.testing.assert_array_equal(cnt, scnt)
        np.testing.assert_array_equal(idx, sidx)
        np.testing.assert_array_equal(dist, sdist)
        np.testing.assert_array_equal(gy, sgy)


@pytest.mark.parametrize("i8", [False, True], ids=["bf16-ring", "int8-ring"])
def test_symmetric_ring_second_round_settles_duplicate_groups(oracle_lib, i8):
    rng = np.random.default_rng(3)
    n, d, k = 2400, 64, 8
    X = clustered(n, d, nclust=12, seed=17)
    for g in range(6):
        rows = rng.choice(n, 150, replace=False)
        X[rows] = X[rows[0]]
    gp = {"eps": calibrate_eps(clustered(n, d, nclust=12, seed=17), k), "k": k, "topk": 5, "p": 2.0, "sigma": None}
    ref = oracle_lib.OracleIndex(X, gp)
    cuts = [0, 800, 1700, 2400]
    res, flagged = _symmetric_ring_lists(X, gp, cuts, i8=i8)   # (int8: the band pass gathers the flagged rows from the image)
    assert flagged > 0
    for b in range(3):
        lo, hi = cuts[b], cuts[b + 1]
        np.testing.assert_array_equal(res[b][3], ref.knn_cnt[lo:hi])
        np.testing.assert_array_equal(res[b][0], ref.knn_idx[lo:hi])


def test_symmetric_ring_with_an_eps_that_admits_every_pair():
    """Only the thresholds gate the visiting items' candidates: buffers that overflow fail their rows' proofs (bound
    -inf) and the second round settles them."""
    n, d, k = 3000, 48, 8
    X = clustered(n, d, nclust=6, seed=5)
    gp = {"eps": 10.0, "k": k, "topk": 5, "p": 2.0, "sigma": None}
    cuts = [0, 900, 2100, 3000]
    res, _ = _symmetric_ring_lists(X, gp, cuts)
    for b in range(3):
        lo, hi = cuts[b], cuts[b + 1]
        sidx, sdist, sgy, scnt = _single_lists(X, gp, lo, hi)
        np.testing.assert_array_equal(res[b][3], scnt)
        np.testing.assert_array_equal(res[b][0], sidx)
        np.testing.assert_array_equal(res[b][1], sdist)


def test_symmetric_ring_seeded_fuzz():
    """Random shapes, cuts (2 to 6 blocks, some tiny), k, metric and eps tightness: the symmetric ring's lists against
    one space holding everything, bit for bit."""
    rng = np.random.default_rng(20261004)
    for case in range(14):
        n = int(rng.integers(600, 5200))
        d = int(rng.choice([16, 40, 96, 200]))
        k = int(rng.integers(3, 31))
        G = int(rng.integers(2, 7))
        inner = np.sort(rng.choice(np.arange(1, n), size=G - 1, replace=False))
        if rng.random() < 0.3:
            inner[0] = max(1, min(int(inner[0]), int(rng.integers(1, 4))))      # a block of a few rows at the front
            inner = np.unique(inner)
        cuts = [0] + [int(v) for v in inner] + [n]
        metric = str(rng.choice(["l2", "cosine"]))
        X = clustered(n, d, nclust=int(rng.integers(2, 12)), noise=float(rng.uniform(0.1, 0.6)), seed=int(rng.integers(1 << 30)))
        eps = calibrate_eps(X, k, metric) * float(rng.choice([0.6, 1.0, 1.0, 2.5]))
        gp = {"eps": eps, "k": k, "topk": 5, "p": 2.0, "sigma": None, "metric": metric}
        res, _ = _symmetric_ring_lists(X, gp, cuts)
        for b in range(len(cuts) - 1):
            lo, hi = cuts[b], cuts[b + 1]
            sidx, sdist, sgy, scnt = _single_lists(X, gp, lo, hi)
            ctx = "case %d: n=%d d=%d k=%d cuts=%s %s eps=%.4g block %d" % (case, n, d, k, cuts, metric, eps, b)
            np.testing.assert_array_equal(res[b][3], scnt, err_msg=ctx)
            np.testing.assert_array_equal(res[b][0], sidx, err_msg=ctx)
            np.testing.assert_array_equal(res[b][1], sdist, err_msg=ctx)


@pytest.mark.parametrize("metric", ["l2", "cosine"])
def test_third_round_settles_thousands_of_duplicates(oracle_lib, metric):
    """5 000 exact copies of one item inside one block, 500 more spread over all blocks, 700 copies of a second item: the
    band of such a row holds more entries than the collection buffers take (4096) in the middle block, the second round
    cannot settle it (overflow), and the third round evaluates every pair exactly -- lists equal to the oracle's (lowest
    ids first among equals) and to the single-space build's."""
    rng = np.random.default_rng(11)
    n, d, k = 9000, 24, 9
    X = clustered(n, d, nclust=10, seed=23, normalise=False)
    item = X[7].copy()
    X[2100:7100] = item
    X[rng.choice(n, 500, replace=False)] = item
    X[rng.choice(n, 700, replace=False)] = X[11].copy()
    gp = {"eps": calibrate_eps(clustered(n, d, nclust=10, seed=23, normalise=False), k, metric), "k": k, "topk": 5, "p": 2.0, "sigma": None,
          "metric": metric}
    ref = oracle_lib.OracleIndex(X, gp)
    cuts = [0, 2000, 7500, 9000]
    for b in range(3):
        lo, hi = cuts[b], cuts[b + 1]
        idx, key, dist, gy, cnt, nflag, over = _ring_lists(X, gp, cuts, b)
        assert nflag > 0 and over > 0
        np.testing.assert_array_equal(cnt, ref.knn_cnt[lo:hi])
        np.testing.assert_array_equal(idx, ref.knn_idx[lo:hi])
        sidx, sdist, sgy, scnt = _single_lists(X, gp, lo, hi)
        np.testing.assert_array_equal(idx, sidx)
        # (the single-space last resort sums a pair's columns in another order than the ring's: last-bit differences)
        np.testing.assert_allclose(dist, sdist, rtol=1e-14, atol=0)
    res, flagged = _symmetric_ring_lists(X, gp, cuts)
    assert flagged > 0
    for b in range(3):
        lo, hi = cuts[b], cuts[b + 1]
        np.testing.assert_array_equal(res[b][3], ref.knn_cnt[lo:hi])
        np.testing.assert_array_equal(res[b][0], ref.knn_idx[lo:hi])


def test_third_round_on_one_dimensional_items():
    """d = 1 (the fuzz case that found the gap): normalised 1-D items are +1 or -1 -- two groups of thousands of
    duplicates; under cosine every pair of a group is at distance 0."""
    import torch

    from pyarrowspace_amd import ArrowSpaceBuilder
    from pyarrowspace_amd.dist import ShardedIndex
    for metric, kernel in (("l2", "gaussian"), ("cosine", "rational")):
        X = clustered(6000, 1, nclust=3, seed=5)
        gp = {"eps": 1e-12, "k": 21, "topk": 10, "p": 2.0, "sigma": None, "metric": metric, "kernel": kernel}
        aspace, gl = ArrowSpaceBuilder.build(gp, X)
        index = ShardedIndex.build(gp, torch.from_numpy(X).cuda())
        assert index.ring_overflowed > 0
        np.testing.assert_allclose(index.lambdas(), np.asarray(aspace.lambdas()), rtol=1e-12)
        q = X[3].copy()
        assert index.search(q, 0.62) == aspace.search(q, gl, 0.62)
        index.close()


class _pair_chunks:
    """as_knn_block_pair takes the visiting block in chunks of `tiles` column tiles (128 items each): what a shard of
    millions of items gets by itself (the transposed buffers of a chunk must fit in an eighth of the free memory)."""

    def __init__(self, tiles):
        self.tiles = tiles

    def __enter__(self):
        import os
        self.old = os.environ.get("ARROWSPACE_PAIR_CHUNK_TILES")
        os.environ["ARROWSPACE_PAIR_CHUNK_TILES"] = str(self.tiles)

    def __exit__(self, *exc):
        import os
        os.environ.pop("ARROWSPACE_PAIR_CHUNK_TILES", None)
        if self.old is not None:
            os.environ["ARROWSPACE_PAIR_CHUNK_TILES"] = self.old


@pytest.mark.parametrize("metric", ["l2", "cosine"])
@pytest.mark.parametrize("n,d,k,cuts,tiles", [(5000, 768, 25, [0, 2500, 5000], 8), (4100, 64, 12, [0, 300, 1400, 2000, 3300, 4100], 3),
                                              (6000, 96, 10, [0, 1700, 4100, 6000], 1)])
def test_symmetric_ring_in_column_chunks(metric, n, d, k, cuts, tiles):
    """The visiting block in chunks of 8, 3 and 1 column tiles: the same lists as one space holding everything, and the
    same as the unchunked ring, bit for bit (a visiting item's slice comes from its one chunk; the own rows' slices of
    the chunks are folded: the M smallest exact keys of their union, the smallest drop bound)."""
    X = clustered(n, d, nclust=max(4, n // 200), seed=n + k + 1)
    gp = {"eps": calibrate_eps(X, k, metric), "k": k, "topk": 5, "p": 2.0, "sigma": None, "metric": metric}
    with _pair_chunks(tiles):
        res, _ = _symmetric_ring_lists(X, gp, cuts)
    plain, _ = _symmetric_ring_lists(X, gp, cuts)
    for b in range(len(cuts) - 1):
        lo, hi = cuts[b], cuts[b + 1]
        single = _single_lists(X, gp, lo, hi)
        for t in range(4):
            np.testing.assert_array_equal(res[b][t], single[t])
            np.testing.assert_array_equal(res[b][t], plain[b][t])


def test_symmetric_ring_in_column_chunks_with_duplicates_and_wide_eps(oracle_lib):
    """Chunks of 2 tiles with duplicate groups (second round) and with an eps that admits every pair."""
    rng = np.random.default_rng(5)
    n, d, k = 2400, 64, 8
    X = clustered(n, d, nclust=12, seed=19)
    for g in range(6):
        rows = rng.choice(n, 150, replace=False)
        X[rows] = X[rows[0]]
    gp = {"eps": calibrate_eps(clustered(n, d, nclust=12, seed=19), k), "k": k, "topk": 5, "p": 2.0, "sigma": None}
    ref = oracle_lib.OracleIndex(X, gp)
    cuts = [0, 800, 1700, 2400]
    with _pair_chunks(2):
        res, flagged = _symmetric_ring_lists(X, gp, cuts)
    assert flagged > 0
    for b in range(3):
        lo, hi = cuts[b], cuts[b + 1]
        np.testing.assert_array_equal(res[b][3], ref.knn_cnt[lo:hi])
        np.testing.assert_array_equal(res[b][0], ref.knn_idx[lo:hi])
    X = clustered(3000, 48, nclust=6, seed=6)
    gp = {"eps": 10.0, "k": k, "topk": 5, "p": 2.0, "sigma": None}
    cuts = [0, 900, 2100, 3000]
    with _pair_chunks(2):
        res, _ = _symmetric_ring_lists(X, gp, cuts)
    for b in range(3):
        sidx, sdist, sgy, scnt = _single_lists(X, gp, cuts[b], cuts[b + 1])
        np.testing.assert_array_equal(res[b][3], scnt)
        np.testing.assert_array_equal(res[b][0], sidx)
        np.testing.assert_array_equal(res[b][1], sdist)


def test_block_pair_refuses_bad_arguments():
    """as_knn_block_pair: a block is not paired with itself, the row range must lie inside the space, the visiting block
    must match the space (features); as_knn_thresholds: row range inside the space."""
    import ctypes as C

    import torch

    from pyarrowspace_amd.dist import HipEngine
    X = clustered(700, 32, nclust=4, seed=1)
    gp = {"eps": calibrate_eps(X, 5), "k": 5, "topk": 5, "p": 2.0, "sigma": None}
    e = HipEngine(gp)
    e.create_space(torch.from_numpy(X[:400].copy()).cuda())
    e.ring_begin(2)
    with pytest.raises(ValueError, match="not paired with itself"):
        e.knn_block_pair(e.own_block(), 0, 400, -1, -1, 0, 0, None, 400)
    h = e.open_block(torch.from_numpy(X[400:].copy()).cuda())
    with pytest.raises(ValueError, match="bad row range"):
        e.knn_block_pair(h, 0, 401, -1, -1, 0, 400, None, 300)
    other = e.open_block(torch.from_numpy(np.ascontiguousarray(X[400:, :16])).cuda())
    with pytest.raises(ValueError, match="does not match"):
        e.knn_block_pair(other, 0, 400, -1, -1, 0, 400, None, 300)
    out = torch.zeros(400, dtype=torch.float32, device="cuda")
    st = e.L.as_knn_thresholds(e.sp, C.byref(e.gp), 0, 401, 1.0, C.c_void_p(e.p_key[0].data_ptr()), C.c_void_p(e.p_cnt[0].data_ptr()),
                               C.c_void_p(out.data_ptr()))
    assert st != 0
    # and a proper call still works afterwards: the empty own list gives +inf thresholds
    assert bool(torch.isinf(e.knn_thresholds(1.0)).all())
    e.close_block(other)
    e.close_block(h)
    e.close()
