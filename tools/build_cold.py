import sys, time, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
import bench, pyarrowspace_amd as asp
dev = torch.device("cuda:0")
X = bench.make_data(1000000, 768, 42, dev)
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a, g = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", 1000000, 768, 768)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    st = g.build_stats()
    print("rep %d: wall %.3f s; ingest %.3f knn %.3f refine %.3f graph %.3f total %.3f" % (rep, dt, st["ingest_s"], st["knn_mfma_s"], st["refine_s"], st["graph_s"], st["total_s"]), flush=True)
    del a, g
