#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel stats / PMC counter collection) into the small
summaries committed under profiles/.  Usage:
    python profiles/summarise.py stats <dir> <out.csv>
    python profiles/summarise.py pmc <dir> <counter> <out.csv>
    python profiles/summarise.py traffic <fetch.csv> <write.csv> <n> <d> <out.json>
    python profiles/summarise.py timeline <dir> <out.csv>
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


def timeline(d, out):
    """Single-query search chain from the kernel trace: per kernel of the chain its average duration and the
    average idle gap in front of it (previous kernel's end -> its start), and the gap between two queries
    (last kernel's end -> next query's first kernel): where the whole-query time beyond the scan goes."""
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r.get("Grid_Size_Z", r.get("Grid_Size_z", 1)) or 1))
            for r in csv.DictReader(open(f)) if "as::" in r["Kernel_Name"]]
    rows.sort()
    # a query = q_prepare (one slot) ... up to the next q_prepare; with a host-prepared query the chain opens with the
    # scan itself (a single-slot scan that no staging kernel precedes)
    chains, cur = [], None
    for st, en, name, gz in rows:
        opens = name.startswith("as::q_prepare_kernel") or (
            (name.startswith("as::scan_dma_kernel") or name.startswith("as::scan_tile_kernel")) and not (cur and len(cur) == 1 and cur[0][2].startswith("as::q_prepare_kernel")))
        if opens:
            if cur:
                chains.append(cur)
            cur = [(st, en, name, gz)]
        elif cur is not None:
            cur.append((st, en, name, gz))
    if cur:
        chains.append(cur)
    single = [c for c in chains if all(g == 1 for _, _, _, g in c) and any("scan_d" in n or "scan_t" in n for _, _, n, _ in c)]
    from collections import Counter
    shape = Counter(tuple(n for _, _, n, _ in c) for c in single).most_common(1)[0][0]
    sel = [c for c in single if tuple(n for _, _, n, _ in c) == shape]
    with open(out, "w", newline="") as fh:
        w = csv.writer(fh)
        w.writerow(["position", "kernel", "avg_us", "avg_gap_before_us", "queries", "p10_us", "p50_us", "p90_us"])
        tot = 0.0
        for i, name in enumerate(shape):
            dur = sum(c[i][1] - c[i][0] for c in sel) / len(sel) / 1e3
            gap = sum((c[i][0] - c[i - 1][1]) for c in sel) / len(sel) / 1e3 if i else 0.0
            tot += dur + gap
            ds = sorted((c[i][1] - c[i][0]) / 1e3 for c in sel)
            w.writerow([i, name, "%.2f" % dur, "%.2f" % gap, len(sel)] + ["%.2f" % ds[int(f_ * (len(ds) - 1))] for f_ in (0.1, 0.5, 0.9)])
        inter = [b[0][0] - a[-1][1] for a, b in zip(sel, sel[1:]) if 0 < b[0][0] - a[-1][1] < 5e6]
        span = sum(c[-1][1] - c[0][0] for c in sel) / len(sel) / 1e3
        w.writerow(["", "device span of one query (first start -> last end)", "%.2f" % span, "", len(sel)])
        w.writerow(["", "idle between queries (host turnaround)", "%.2f" % (sum(inter) / max(len(inter), 1) / 1e3), "", len(inter)])
    print(open(out).read())


def traffic(fetch_csv, write_csv, n, d, out):
    """HBM-side bytes per launch of the roofline kernels, corrected as MI355X_MICROARCH.md prescribes for
    gfx950: FETCH_SIZE (KiB) x 2 for 16-B-per-lane streaming reads, WRITE_SIZE (KiB) as read."""
    import json

    def per_dispatch(path):
        return {r["kernel"]: float(r[[k for k in r if k.endswith("_per_dispatch")][0]]) for r in csv.DictReader(open(path))}

    f, w = per_dispatch(fetch_csv), per_dispatch(write_csv)
    doc = {"_comment": traffic.__doc__.strip().replace("\n    ", " ") + "  Separate --pmc passes (profiles/*_pmc_*.csv). "
                       "Valid only for the workload named in 'workload'.",
           "workload": {"n": int(n), "d": int(d)}}
    for key, prefix in (("scan_tile_kernel", "as::scan_tile_kernel"), ("scan_dma_kernel", "as::scan_dma_kernel"), ("scan_dots_f32_kernel", "as::scan_dots_f32_kernel"),
                        ("scan_gemm_kernel", "as::scan_gemm_kernel"), ("scan_gemm_dual_kernel", "as::scan_gemm_dual_kernel"), ("knn_mfma_kernel", "as::knn_mfma"),
                        ("knn_bf16_kernel", "as::knn_bf16_kernel")):
        fk = [k for k in f if k.startswith(prefix)]
        if not fk:
            continue
        k = max(fk, key=lambda name: f[name])   # (the main pass of the build, not its threshold pass)
        doc[key] = {"kernel": k, "fetch_kib": f[k], "write_kib": w.get(k, 0.0), "bytes_per_launch": int((2.0 * f[k] + w.get(k, 0.0)) * 1024)}
    if "knn_bf16_kernel" in doc:
        doc["knn_bf16_kernel"]["note"] = ("fabric-side requests of the symmetric main pass (bf16 head + tail kernel); Infinity-Cache hits are "
                                          "counted (MI355X_MICROARCH.md HBM section)")
    if "knn_mfma_kernel" in doc:
        doc["knn_mfma_kernel"]["note"] = ("fabric-side requests of the 8-wave LDS-DMA kernel; Infinity-Cache hits are counted "
                                          "(MI355X_MICROARCH.md HBM section)")
    json.dump(doc, open(out, "w"), indent=1)
    print(open(out).read())


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    elif sys.argv[1] == "traffic":
        traffic(*sys.argv[2:7])
    elif sys.argv[1] == "timeline":
        timeline(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4])
