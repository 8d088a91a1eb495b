#!/bin/bash
# Runs on the GPU box: the batched pass over the tau range (small tau: the lambda term decides, and the coarse keys keep it in fp64).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT" || exit 1
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "single %.1f q/s" % d["value"], "batched %.0f q/s" % d["batched_queries_per_sec"], "frac %.3f" % d["roofline_batch"]["frac"])'
for T in 0.0 0.03 0.05 0.2 0.62 1.0; do
  python bench.py --no-cpu-baseline --no-live-traffic --steps 50 --warmup 5 --tau $T 2>/dev/null | python -c "$J" "tau $T"
done
