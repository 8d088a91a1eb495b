#!/usr/bin/env python3
"""Ramp and tail of the single query's tile scan (make STAMPS=1 build): per wave, when it started and when it ended (100 MHz
wall clock), over a few queries -- how long after the first wave the last one starts, how long before the last wave's end the
median wave ends.   python tools/scan_stamps.py [N] [D]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import pyarrowspace_amd as asp
from pyarrowspace_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
dev = torch.device("cuda:0")
X = bench.make_data(n, d, 42, dev)
Q = bench.make_queries(X, 64, 43)
gp = {"eps": bench.calibrate_eps(X, 25), "k": 25, "topk": 15, "p": 2.0, "sigma": None}
aspace, gl = asp.ArrowSpaceBuilder.build_from_device(gp, X.data_ptr(), "float32", n, d, d)
L = _lib.load()
fn = L.as_debug_scan_stamps
fn.restype = C.c_int
buf = np.zeros(2 * 4096, dtype=np.uint64)
for i in range(20):
    aspace.search(Q[i], gl, 0.62)
rows = []
for i in range(20, 40):
    aspace.search(Q[i], gl, 0.62)
    torch.cuda.synchronize()
    assert fn(buf.ctypes.data_as(C.c_void_p)) == 0
    st, en = buf[:4096].astype(np.int64), buf[4096:].astype(np.int64)
    live = en > 0
    st, en = st[live], en[live]
    t0 = st.min()
    s_us, e_us = (st - t0) / 100.0, (en - t0) / 100.0
    rows.append((live.sum(), s_us.max(), np.percentile(s_us, 50), e_us.max(), np.percentile(e_us, 50), np.percentile(e_us, 10), np.percentile(e_us, 90), (e_us - s_us).mean()))
r = np.array(rows, dtype=np.float64)
print("waves %d: last start %.1f us after the first (median %.1f); ends: p10 %.1f p50 %.1f p90 %.1f last %.1f us; mean wave lifetime %.1f us"
      % (r[:, 0].mean(), r[:, 1].mean(), r[:, 2].mean(), r[:, 5].mean(), r[:, 4].mean(), r[:, 6].mean(), r[:, 3].mean(), r[:, 7].mean()))
