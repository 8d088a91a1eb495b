#!/usr/bin/env python3
"""Launch geometry of the tile scan (single query, coarse operand) on ONE built index: queries/s and the scan's duration per
geometry <blocks per CU><ring KiB per wave>.   python tools/tile_geom.py [N] [D] [geom ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pyarrowspace_amd as asp
from pyarrowspace_amd import _lib


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
    geoms = [int(a) for a in sys.argv[3:]] or [406, 306, 206, 308, 208, 212, 216, 116]
    slacks = [int(v) for v in os.environ.get("TILE_DYN", "1").split(",")]
    dev = torch.device("cuda:0")
    X = bench.make_data(n, d, 42, dev)
    Q = bench.make_queries(X, 256, 43)
    gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
    aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, d, d)
    L = _lib.load()
    d8 = (d + 63) // 64 * 64
    for rep in range(2):
        for g, sl in [(g, sl) for g in geoms for sl in slacks]:
            L.as_set_tuning(b"tile_geom", g)
            L.as_set_tuning(b"tile_dyn", sl)
            for i in range(20):
                aspace.search(Q[i], gl, 0.62)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for i in range(200):
                aspace.search(Q[20 + i], gl, 0.62)
            dt = time.perf_counter() - t
            asp.enable_search_stats(True)
            us = []
            for i in range(50):
                aspace.search(Q[i], gl, 0.62)
                us.append(aspace.last_search_stats()["scan_us"])
            asp.enable_search_stats(False)
            print(f"geom={g} dyn={sl} n={n} d={d}: {200 / dt:.0f} q/s, scan {np.mean(us):.1f} us (min {np.min(us):.1f}) = {n * (d8 + 12) / np.mean(us) / 1e6:.2f} TB/s moved, operand {aspace.last_scan_operand}", flush=True)


if __name__ == "__main__":
    main()
