"""The ring build's block passes on the bf16 head + tail images against the shards' int8 images: two blocks of N / 2 rows in one
process (own-block passes, one pair pass, merge), wall seconds of the k-NN part.  python tools/ring_pipe_ab.py [N] [D]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from pyarrowspace_amd.dist import HipEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
X = bench.make_data(n, d, 42, torch.device("cuda", 0))
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
cuts = [0, n // 2, n]
blocks = [X[cuts[b]:cuts[b + 1]].contiguous() for b in range(2)]
ref = None
for i8 in (False, True, False, True):
    eng = []
    for b in range(2):
        e = HipEngine(gp)
        e.create_space(blocks[b])
        e.ring_begin(2)
        eng.append(e)
    nmax = [e.block_nmax(e.own_block()) for e in eng]
    if i8:
        st = np.array([e.ring_i8_stats() for e in eng])
        assert all(e.ring_i8_set(st[:, 0].max(), st[:, 1].max(), st[:, 2].max() == 0.0) for e in eng)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for r, e in enumerate(eng):
        e.knn_block(e.own_block(), r, cuts[r], cuts[r])
    U = [e.knn_thresholds(max(nmax)) for e in eng]
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    counts = [cuts[1] - cuts[0], cuts[2] - cuts[1]]
    # world 2: the opposite pair is split -- rank q = 1 takes the rows of its second half of tiles, rank 0 the first half of the columns
    out = {}
    for r, e in enumerate(eng):
        src = (r - 1) % 2
        row0, row1, ct0, ct1 = 0, counts[r], -1, -1
        q = max(r, src)
        tq = (counts[q] + 255) // 256 * 256 // 128
        if r == q:
            row0 = min(counts[q], (tq // 2) * 128)
        else:
            ct0, ct1 = 0, tq // 2
        h = e.open_block(blocks[src])
        out[(r, src)] = e.knn_block_pair(h, row0, row1, ct0, ct1, cuts[r], cuts[src], U[src], counts[src], row_thr=U[r])
        e.close_block(h)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for (r, src), P in out.items():
        eng[src].fold_slice(P, nmax[r])
    flagged = sum(e.knn_merge(nmax) for e in eng)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    lists = [[t.cpu().numpy() for t in e.lists()] for e in eng]
    if ref is None:
        ref = lists
    same = all(np.array_equal(a, b) for la, lb in zip(lists, ref) for a, b in zip(la, lb))
    print("%s ring: own blocks %.3f s, pair passes %.3f s, fold + merge %.3f s, flagged rows %d, lists equal to the first run's: %s"
          % ("int8" if i8 else "bf16", t1 - t0, t2 - t1, t3 - t2, flagged, same))
    for e in eng:
        e.close()
