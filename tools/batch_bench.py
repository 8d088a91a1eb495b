"""Batched search throughput (as_search_batch) on a synthetic index: python tools/batch_bench.py [N] [D] [B]."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pyarrowspace_amd as asp
import bench

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    dev = torch.device("cuda:0")
    X = bench.make_data(n, d, 42, dev)
    gq = torch.Generator(device=dev)
    gq.manual_seed(43)
    qrows = torch.randint(0, n, (B,), generator=gq, device=dev)
    Qd = X[qrows] + 0.025 / 31.0 * torch.randn((B, d), generator=gq, device=dev, dtype=torch.float32)
    Q = (Qd / Qd.norm(dim=1, keepdim=True)).double().cpu().numpy()
    gp = {"eps": bench.calibrate_eps(X, 16), "k": 16, "topk": 10, "p": 2.0, "sigma": None}
    aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, d, d)
    Q = np.ascontiguousarray(Q[:B])
    if os.environ.get("BATCH_BENCH_NOCHECK"):      # diagnostic kernel variants return garbage: time only
        for _ in range(6):
            try:
                aspace.search_batch(Q, gl, 0.62)
            except BaseException as e:
                print("ignored:", type(e).__name__)
        return
    got = aspace.search_batch(Q, gl, 0.62)
    for b in (0, 7, 31, 32, B - 1):
        assert got[b] == aspace.search(Q[b], gl, 0.62), b
    torch.cuda.synchronize()
    ts = []
    for _ in range(12):
        t = time.perf_counter()
        aspace.search_batch(Q, gl, 0.62)
        ts.append(time.perf_counter() - t)
    dt = float(np.median(ts))
    print(f"variant={os.environ.get('ARROWSPACE_GEMM_VARIANT','0')} n={n} d={d} B={B}: {B/dt:.0f} q/s median "
          f"({dt/((B+31)//32)*1e3:.3f} ms per 32-slot pass; calls {min(ts)*1e3:.1f}..{max(ts)*1e3:.1f} ms)")


if __name__ == "__main__":
    main()
