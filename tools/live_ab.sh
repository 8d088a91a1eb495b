#!/bin/bash
# Runs on the GPU box: does the live traffic probe (two rocprofv3 --pmc child passes in front) change what the run then measures?
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], "%.1f q/s" % d["value"], "scan %.1f us" % (d["roofline"]["avg_launch_ms"]*1e3), "traffic", d["roofline"]["traffic"], "build %.2f s" % d["index_build_sec"])'
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "$J" "no probe  "
  python bench.py --no-cpu-baseline 2>/dev/null | python -c "$J" "live probe"
done
python bench.py --no-cpu-baseline --no-live-traffic 2>/dev/null | python -c "$J" "no probe  "
