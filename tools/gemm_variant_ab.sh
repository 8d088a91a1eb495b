#!/bin/bash
# Runs on the GPU box: the batched scan with a ring of 4 slabs (default) against 3 (ARROWSPACE_GEMM_VARIANT=1: 61.5 KB of LDS per block,
# 37 KB of every CU left to the other workspace's kernels), now that the two workspaces overlap.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT" || exit 1
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "%.0f q/s" % d["batched_queries_per_sec"], "%.4f ms/pass" % d["roofline_batch"]["ms_per_pass"], "frac %.3f" % d["roofline_batch"]["frac"])'
for V in 0 1 0 1; do
  ARROWSPACE_GEMM_VARIANT=$V python bench.py --no-cpu-baseline --no-live-traffic --steps 50 --warmup 5 2>/dev/null | python -c "$J" "variant $V"
done
