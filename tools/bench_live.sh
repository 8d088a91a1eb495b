#!/bin/bash
# Runs on the GPU box: the default bench.py (live PMC traffic passes + CPU baseline), wall time of the whole command, the roofline objects.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
mkdir -p gpurun_out
S=$(date +%s)
python bench.py "$@" > gpurun_out/bench_live.json 2> gpurun_out/bench_live.err || { tail -20 gpurun_out/bench_live.err; exit 1; }
echo "whole command: $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/bench_live.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"])
print("batch traffic", d["roofline_batch"]["traffic"], "build traffic", d["roofline_build"]["traffic"])
PY
