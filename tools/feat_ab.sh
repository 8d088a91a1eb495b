#!/bin/bash
# Runs on the GPU box: feature mode (cosine / rational, lambda on the F x F Laplacian), fused tail against round 2's chain, alternating.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT" || exit 1
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "%.1f q/s" % d["value"], "%.4f ms/step" % d["ms_per_step"], "query frac %.4f" % d["roofline_query"]["frac"], "in-dist %.1f q/s" % d["in_distribution_queries"]["value"])'
A="--no-cpu-baseline --no-live-traffic --steps 400 --lambda-mode feature --metric cosine --kernel rational"
for i in 1 2; do
  ARROWSPACE_NO_FUSED_TAIL=1 python bench.py $A "$@" 2>/dev/null | python -c "$J" "chain (5 launches)"
  python bench.py $A "$@" 2>/dev/null | python -c "$J" "fused (3 launches)"
done
