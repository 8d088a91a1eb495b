#!/bin/bash
# Runs on the GPU box: does the timed region need the 300 priming searches?  Staged path (library-side exchange, one rank) and fused path, with and without.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT" || exit 1
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "primed", d["priming_queries"], "%.1f q/s" % d["value"], "%.4f ms/step" % d["ms_per_step"], "in-dist %.1f q/s" % d["in_distribution_queries"]["value"])'
for P in 0 300 0 300; do
  ARROWSPACE_BENCH_PRIME=$P ARROWSPACE_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-live-traffic --steps 400 "$@" 2>/dev/null | python -c "$J" "staged (library RCCL)"
done
for P in 0 300; do
  ARROWSPACE_BENCH_PRIME=$P python bench.py --no-cpu-baseline --no-live-traffic --steps 400 "$@" 2>/dev/null | python -c "$J" "fused (one GPU)      "
done
