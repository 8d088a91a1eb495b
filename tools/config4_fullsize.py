"""BASELINE.json config 4 at its full size as a real 4-rank job on ONE GPU: 8.8M x 768 fp32 row-sharded over 4 processes
(2.2M rows each), the whole protocol of DESIGN.md section 6 (tests/test_gpu_multirank.py's worker: ring with every pair
of shards once and the split pair, slices home, edge all-to-all, sharded graph stage, staged single + batched search
with the tau sweep of tests/test_4_msmarco_tau_sweep.py), the ranks sharing the card (exchange steps staged through
host memory).  No single-space build to compare with at this size: Laplacian identities over all ranks, sampled rows
against an fp64 brute force, scores from the definition, single == batched on all ranks.
usage: config4_fullsize.py [N] [ranks] [uneven] [metric kernel]
(5 ranks, uneven, cosine rational at 8M: the protocol of config 5 -- odd ring, shards of 65-135 % of the mean -- at the
largest size one GPU finishes in minutes)"""
import os, sys, threading, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_multirank as m

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_800_000
world = int(sys.argv[2]) if len(sys.argv) > 2 else 4
uneven = len(sys.argv) > 3 and sys.argv[3] == "uneven"
metric, kernel = (sys.argv[4], sys.argv[5]) if len(sys.argv) > 5 else ("l2", "gaussian")


def heartbeat():
    t0 = time.time()
    while True:
        time.sleep(30)
        print("  ... %.0f s" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    threading.Thread(target=heartbeat, daemon=True).start()
    t0 = time.time()
    m._run(world, n, 768, uneven=uneven, metric=metric, kernel=kernel, single=False)
    print("N = %d x 768 on %d ranks (%s shards, %s / %s) sharing one GPU: %.0f s in all"
          % (n, world, "uneven" if uneven else "even", metric, kernel, time.time() - t0), flush=True)
