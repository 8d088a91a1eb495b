#!/usr/bin/env python3
"""Timeline of as_search_batch's kernels from a rocprofv3 --kernel-trace run of tools/batch_bench.py:
   python tools/batch_trace.py <dir with *kernel_trace.csv> [first scan to print from]"""
import csv
import glob
import os
import sys

f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True))[-1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "")[:58], r.get("Stream_Id")) for r in csv.DictReader(open(f)))
idx = [i for i, k in enumerate(ks) if "scan_gemm" in k[2]]
i0 = idx[min(skip, len(idx) - 1)]
t0 = ks[i0][0]
for k in ks[max(0, i0 - 4):i0 + 70]:
    print("%9.1f %9.1f %7.1f  s=%s %s" % ((k[0] - t0) / 1e3, (k[1] - t0) / 1e3, (k[1] - k[0]) / 1e3, k[3], k[2]))
