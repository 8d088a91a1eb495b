#!/bin/bash
# Runs on the GPU box: batched passes alternating two workspaces (default) against one workspace, same box, bench's configuration.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT" || exit 1
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "%.0f q/s" % d["batched_queries_per_sec"], "%.4f ms/pass" % d["roofline_batch"]["ms_per_pass"], "frac %.3f" % d["roofline_batch"]["frac"])'
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-live-traffic --steps 20 --warmup 2 2>/dev/null | python -c "$J" "two workspaces"
  ARROWSPACE_NO_BATCH_PIPELINE=1 python bench.py --no-cpu-baseline --no-live-traffic --steps 20 --warmup 2 2>/dev/null | python -c "$J" "one workspace "
done
