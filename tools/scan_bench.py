"""Single-query scan timing on a synthetic index: python tools/scan_bench.py [N] [D] -> q/s and scan kernel ms."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pyarrowspace_amd as asp
import bench

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    dev = torch.device("cuda:0")
    X = bench.make_data(n, d, 42, dev)
    gq = torch.Generator(device=dev); gq.manual_seed(43)
    qrows = torch.randint(0, n, (64,), generator=gq, device=dev)
    Qd = X[qrows] + 0.025 / 31.0 * torch.randn((64, d), generator=gq, device=dev, dtype=torch.float32)
    Q = (Qd / Qd.norm(dim=1, keepdim=True)).double().cpu().numpy()
    gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
    aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, d, d)
    for i in range(20): aspace.search(Q[i % 64], gl, 0.62)
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(200): aspace.search(Q[i % 64], gl, 0.62)
    dt = time.perf_counter() - t
    asp.enable_search_stats(True)
    us = []
    for i in range(50):
        aspace.search(Q[i % 64], gl, 0.62); us.append(aspace.last_search_stats()["scan_us"])
    print(f"variant={os.environ.get('ARROWSPACE_SCAN_VARIANT','0')} n={n} d={d}: {200/dt:.0f} q/s, scan {np.mean(us):.1f} us = {n*(d+1)*4/np.mean(us)/1e6:.2f} TB/s")

if __name__ == "__main__":
    main()
