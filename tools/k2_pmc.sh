#!/bin/bash
# PMC passes over one index build (tools/build_only.py N): where the bf16 K2 kernel's time goes.
# usage: bash tools/k2_pmc.sh <tag> [N]        env (ARROWSPACE_*) is passed through
set -o pipefail
TAG=${1:-k2}; N=${2:-262144}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
pass() {  # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 tools/build_only.py $N 1 > $OUT/$name.log 2>&1 || { tail -5 $OUT/$name.log; return 1; }
  python3 - "$OUT/$name" "$@" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True))[0]
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if "knn_bf16" not in k and "knn_mfma" not in k: continue
    k = k.replace("void ", "").split("(")[0][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[(k, r["Counter_Name"])] += 1
for k in acc:
    print(k, {c: "%.4g" % v for c, v in acc[k].items()}, "dispatches", max(cnt[(k, c)] for c in acc[k]))
PY
}
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE &&
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum &&
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE &&
pass sq2 SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM
pass mops8 SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA || echo "(no int8 MFMA op counter under these names on this box)"
grep -h "n=" $OUT/*.log | head -8
