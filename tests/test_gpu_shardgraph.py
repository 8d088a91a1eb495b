"""GPU: the graph stage without replication (SURVEY 8e "Symmetrise + Laplacian", as_graph_shard_csr / _energy /
_lambdas).  The items are cut into uneven shards on ONE device; every shard is its own space and receives exactly the
directed edges whose target row it owns -- what the host's variable-count all-to-all delivers between ranks -- and the
all-gathers are concatenations.  Every shard's rows of the CSR, its degrees, energies and lambdas must equal the
single-space graph's bit for bit (same entries, same ascending-column order, same sums)."""
import ctypes as C

import numpy as np
import pytest

from conftest import calibrate_eps, clustered

pytestmark = pytest.mark.gpu


def _csr(e):
    """(indptr, indices, values) of an engine's graph through as_graph_csr (diagonal included)."""
    L = e.L
    rows, nnz = int(L.as_nnodes(e.gr)), int(L.as_graph_nnz(e.gr))
    ip, ix, v = np.zeros(rows + 1, dtype=np.int64), np.zeros(nnz, dtype=np.int64), np.zeros(nnz)
    e._check(L.as_graph_csr(e.gr, ip.ctypes.data_as(C.c_void_p), ix.ctypes.data_as(C.c_void_p), v.ctypes.data_as(C.c_void_p)))
    return ip, ix, v


def _degrees(e):
    out = np.zeros(int(e.L.as_nnodes(e.gr)))
    e._check(e.L.as_graph_degrees(e.gr, out.ctypes.data_as(C.c_void_p)))
    return out


def _whole(X, gp):
    import torch

    from pyarrowspace_amd.dist import HipEngine
    e = HipEngine(gp)
    e.create_space(torch.from_numpy(X).cuda())
    lists = e.knn_rows(0, X.shape[0])
    e.graph_from_knn(*lists)
    return e, lists, e.norms()


@pytest.mark.parametrize("metric,kernel", [("l2", "gaussian"), ("cosine", "rational")])
@pytest.mark.parametrize("n,d,k,cuts", [(3000, 96, 10, [0, 700, 1900, 3000]), (900, 40, 6, [0, 100, 101, 600, 900])])
def test_shard_rows_equal_the_whole_graph_bitwise(metric, kernel, n, d, k, cuts):
    import torch

    from pyarrowspace_amd.dist import HipEngine
    X = clustered(n, d, nclust=8, seed=3)
    gp = {"eps": calibrate_eps(X, k, metric), "k": k, "topk": 5, "p": 2.0, "sigma": None, "metric": metric, "kernel": kernel}
    whole, (idx, dist, gy, cnt), n64 = _whole(X, gp)
    ip_w, ix_w, v_w = _csr(whole)
    deg_w, lam_w = _degrees(whole), whole.lambdas().copy()
    tau0_w = whole.tau0()
    # every directed edge (src -> tgt) of the whole graph, as the ranks would hold them
    valid = torch.arange(k, device=idx.device)[None, :] < cnt[:, None].long()
    src = torch.arange(n, device=idx.device)[:, None].expand(n, k)[valid]
    tgt, ed, eg = idx[valid].long(), dist[valid], gy[valid]
    shards = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        e = HipEngine(gp)
        e.create_space(torch.from_numpy(X[lo:hi].copy()).cuda())
        here = (tgt >= lo) & (tgt < hi)
        deg = e.graph_shard_csr(n, lo, idx[lo:hi].contiguous(), dist[lo:hi].contiguous(), gy[lo:hi].contiguous(), cnt[lo:hi].contiguous(),
                                (tgt[here] - lo).int(), src[here].int(), ed[here], eg[here])
        shards.append((e, deg))
    deg_g = torch.cat([dg for _, dg in shards]).contiguous()
    np.testing.assert_array_equal(deg_g.cpu().numpy(), deg_w)
    E_g = torch.cat([e.graph_shard_energy(deg_g, n64) for e, _ in shards]).contiguous()
    for (e, _), lo, hi in zip(shards, cuts[:-1], cuts[1:]):
        e.graph_shard_lambdas(E_g)
        assert e.tau0() == tau0_w
        np.testing.assert_array_equal(e.lambdas(), lam_w[lo:hi])
        assert int(e.L.as_graph_row_offset(e.gr)) == lo and int(e.L.as_graph_ncols(e.gr)) == n
        ip, ix, v = _csr(e)
        np.testing.assert_array_equal(ip, ip_w[lo : hi + 1] - ip_w[lo])
        np.testing.assert_array_equal(ix, ix_w[ip_w[lo] : ip_w[hi]])
        np.testing.assert_array_equal(v, v_w[ip_w[lo] : ip_w[hi]])
        e.close()
    whole.close()


def test_shard_graph_refuses_edges_outside_the_shard():
    import torch

    from pyarrowspace_amd.dist import HipEngine
    n, d, k = 600, 32, 5
    X = clustered(n, d, nclust=4, seed=9)
    gp = {"eps": calibrate_eps(X, k), "k": k, "topk": 5, "p": 2.0, "sigma": None}
    whole, (idx, dist, gy, cnt), _ = _whole(X, gp)
    e = HipEngine(gp)
    e.create_space(torch.from_numpy(X[:200].copy()).cuda())
    dev = idx.device
    one = lambda v, dt: torch.tensor([v], dtype=dt, device=dev)   # noqa: E731
    args = (idx[:200].contiguous(), dist[:200].contiguous(), gy[:200].contiguous(), cnt[:200].contiguous())
    with pytest.raises(ValueError, match="outside"):          # target row 200 does not live in a 200-row shard
        e.graph_shard_csr(n, 0, *args, one(200, torch.int32), one(5, torch.int32), one(0.1, torch.float64), one(0.9, torch.float64))
    with pytest.raises(ValueError, match="outside"):          # source item beyond the n items
        e.graph_shard_csr(n, 0, *args, one(3, torch.int32), one(n, torch.int32), one(0.1, torch.float64), one(0.9, torch.float64))
    with pytest.raises(ValueError, match="outside"):          # the shard's rows do not fit under n_global
        e.graph_shard_csr(150, 0, *args, one(3, torch.int32), one(5, torch.int32), one(0.1, torch.float64), one(0.9, torch.float64))
    e.close()
    whole.close()


@pytest.mark.parametrize("gather_lists", [False, True], ids=["sharded_graph", "gathered_lists"])
def test_one_rank_index_equals_the_plain_build_bitwise(gather_lists):
    """ShardedIndex on one rank (ring k-NN over the one block, then either graph stage) against ArrowSpaceBuilder.build."""
    import torch

    from pyarrowspace_amd import ArrowSpaceBuilder
    from pyarrowspace_amd.dist import ShardedIndex
    n, d, k = 4000, 128, 12
    X = clustered(n, d, nclust=8, seed=5)
    gp = {"eps": calibrate_eps(X, k), "k": k, "topk": 8, "p": 2.0, "sigma": None}
    aspace, gl = ArrowSpaceBuilder.build(gp, X)
    index = ShardedIndex.build(gp, torch.from_numpy(X).cuda(), gather_lists=gather_lists)
    np.testing.assert_array_equal(index.lambdas(), np.asarray(aspace.lambdas()))
    rng = np.random.default_rng(2)
    for _ in range(6):
        q = X[rng.integers(0, n)] + 0.02 * rng.standard_normal(d) / np.sqrt(d)
        for tau in (1.0, 0.62):
            assert index.search(q, tau) == aspace.search(q, gl, tau)
    index.close()
