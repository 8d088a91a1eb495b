#!/bin/bash
# Runs on the GPU box: the paired batched pass (one scan for 64 queries) -- grid of the shared scan as a share of 2 blocks per CU
# (above 100: blocks that retire during the scan), the scan on a stream of its own at the lowest priority or on the first workspace's.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT" || exit 1
for cfg in "100 0" "100 1" "400 1" "800 1" "1600 1" "800 0"; do
  set -- $cfg
  echo "== grid $1 % of 2 blocks per CU, scan stream of its own $2"
  ARROWSPACE_BATCH_GRID_PCT=$1 ARROWSPACE_BATCH_SCAN_STREAM=$2 ARROWSPACE_BATCH_ORDER=0 timeout -k 10 200 python tools/batch_bench.py 1000000 768 1024 2>&1 | grep variant
done
