#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel stats / PMC counter collection) into the small
summaries committed under profiles/.  Usage:
    python profiles/summarise.py stats <dir> <out.csv>
    python profiles/summarise.py pmc <dir> <counter> <out.csv>
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0][:80]


def stats(d, out):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))[0]
    rows = [r for r in csv.DictReader(open(f)) if "as::" in r["Name"]]
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct_of_gpu_time"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], "%.3f" % (float(r["TotalDurationNs"]) / 1e6),
                        "%.2f" % (float(r["AverageNs"]) / 1e3), "%.2f" % (float(r["MinNs"]) / 1e3),
                        "%.2f" % (float(r["MaxNs"]) / 1e3), r["Percentage"]])
    print(open(out).read())


def pmc(d, counter, out):
    f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[0]
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != counter or "as::" not in r["Kernel_Name"]:
            continue
        a = acc[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "dispatches", counter + "_sum", counter + "_per_dispatch"])
        for k, (n, v) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
            w.writerow([k, n, "%.1f" % v, "%.1f" % (v / n)])
    print(open(out).read())


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
