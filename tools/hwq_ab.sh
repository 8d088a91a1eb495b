#!/bin/bash
# Runs on the GPU box: do the two batched workspaces' streams share one hardware queue?  GPU_MAX_HW_QUEUES (ROCm: 4 by default).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT" || exit 1
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "%.0f q/s" % d["batched_queries_per_sec"], "%.4f ms/pass" % d["roofline_batch"]["ms_per_pass"], "frac %.3f" % d["roofline_batch"]["frac"], "| single %.1f q/s" % d["value"])'
for Q in 4 8 16 4 8; do
  GPU_MAX_HW_QUEUES=$Q python bench.py --no-cpu-baseline --no-live-traffic --steps 50 --warmup 5 2>/dev/null | python -c "$J" "GPU_MAX_HW_QUEUES=$Q"
done
