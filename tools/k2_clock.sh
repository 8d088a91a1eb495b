#!/bin/bash
# effective clock and matrix-pipe utilisation of the K2 kernels of one build: GRBM_GUI_ACTIVE / 8 / duration, MFMA busy / SIMD cycles
TAG=${1:-clk}; N=${2:-262144}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
OUT=gpurun_out/clk_$TAG; mkdir -p $OUT
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/p -- python3 tools/build_only.py $N 1 > $OUT/p.log 2>&1 || { tail -5 $OUT/p.log; exit 1; }
python3 - $OUT/p <<'PY'
import csv, glob, os, sys
d = sys.argv[1]
cc = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(cc)))
by = {}
for r in rows:
    if "knn_" not in r["Kernel_Name"]: continue
    key = (r["Dispatch_Id"], r["Kernel_Name"].replace("void ", "").split("(")[0][:50])
    by.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    if "Start_Timestamp" in r: by[key]["_dur"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
for (did, name), c in by.items():
    dur = c.get("_dur", 0)
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print("%s dur %.4fs clock %.2f GHz  mfma_busy %.1f%%  wait_any %.1f%% wait_inst %.1f%% active %.1f%%" % (
        name, dur, cyc / dur / 1e9 if dur else 0, 100 * c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc,
        100 * c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 100 * c["SQ_ACTIVE_INST_ANY"] / c["SQ_WAVE_CYCLES"]))
PY
grep -h "n=" $OUT/p.log
