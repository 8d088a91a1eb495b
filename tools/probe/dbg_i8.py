import sys, os
sys.path.insert(0, "/root/repo")
import torch, bench
import pyarrowspace_amd as asp
asp.set_debug(True)
X = bench.make_data(65536, 768, 42, torch.device("cuda", 0))
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
a, g = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", 65536, 768, 768)
