#!/usr/bin/env python3
"""Kernel chain of one query on an unfriendly distribution, from a rocprofv3 --kernel-trace run of tools/dist_sweep.py:
   python tools/dist_trace.py <dir with *kernel_trace.csv>  -- the last 6 queries' kernels, microseconds from the scan's start."""
import csv
import glob
import os
import sys

f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "")[:64]) for r in csv.DictReader(open(f)) if "as::" in r["Kernel_Name"])
scans = [i for i, k in enumerate(ks) if k[2].startswith("as::scan_")]
for si in scans[-7:-1]:
    t0 = ks[si][0]
    nxt = [j for j in scans if j > si][0]
    for k in ks[si:nxt]:
        print("%8.1f %8.1f %7.1f  %s" % ((k[0] - t0) / 1e3, (k[1] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[2]))
    print("   next scan starts at %.1f" % ((ks[nxt][0] - t0) / 1e3))
