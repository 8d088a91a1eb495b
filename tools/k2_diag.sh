#!/bin/bash
# timing skeletons of the bf16 K2 kernel (library built with make ABLATION=1); wrong results by design
# usage: k2_diag.sh N "diag list"
cd "$(dirname "$0")/.."
N=${1:-262144}
for d in ${2:-0 4 5 13 6 12 14}; do
  echo -n "diag=$d  "; ARROWSPACE_K2_DIAG=$d ARROWSPACE_NO_BAND_PASS=1 timeout -k 10 200 python tools/build_only.py $N 1 2>&1 | tail -1
done
