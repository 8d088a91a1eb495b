"""One whole pair pass of the symmetric ring at a large shard shape, on one GPU: rank 0 holds N rows (own block first:
its rows' thresholds seed the pair pass, as on the ring), the visiting shard holds N other rows of the same data;
as_knn_block_pair computes the pair once for both sides.  Reports seconds and TFLOP/s issued (2 N^2 D).
usage: pair_bench.py N [D]       (BASELINE config 5's shard is 8M x 768: a pair there is 4 x the 4M x 4M one)"""
import os, sys, threading, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import torch
import bench
from conftest import gpu_clustered
from pyarrowspace_amd.dist import HipEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
stop = False


def heartbeat():
    t0 = time.time()
    while not stop:
        time.sleep(30)
        print("  ... %.0f s" % (time.time() - t0), flush=True)


threading.Thread(target=heartbeat, daemon=True).start()
X = gpu_clustered(2 * n, d, 42)
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
A, B = X[:n].clone(), X[n:].clone()
del X
torch.cuda.empty_cache()
e = HipEngine(gp)
e.create_space(A)
e.ring_begin(2)
torch.cuda.synchronize()
t0 = time.perf_counter()
e.knn_block(e.own_block(), 0, 0, 0)
torch.cuda.synchronize()
t_own = time.perf_counter() - t0
print("own block %d x %d: %.2f s (symmetric pass: %.1f TFLOP/s of the full square's 2 N^2 D)" % (n, n, t_own, 2.0 * n * n * d / t_own / 1e12), flush=True)
U = e.knn_thresholds(1.0)
h = e.open_block(B)
torch.cuda.synchronize()
t1 = time.perf_counter()
P = e.knn_block_pair(h, 0, n, -1, -1, 0, n, U[:n], n, row_thr=U)
torch.cuda.synchronize()
t_pair = time.perf_counter() - t1
print("pair pass %d x %d: %.2f s = %.1f TFLOP/s issued; both sides' slices from one pass (a rank of an 8-rank ring does 3.5 of these)"
      % (n, n, t_pair, 2.0 * n * n * d / t_pair / 1e12), flush=True)
e.close_block(h)
e.close()
stop = True
