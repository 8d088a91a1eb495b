#!/bin/bash
# one GPU-box session: the -m gpu suite (streamed to a file so a hang shows where), then short benches
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 200 --timeout-method=thread --durations=8 > gpurun_out/r2_tests.log 2>&1; rc=$?
tail -14 gpurun_out/r2_tests.log
[ $rc -eq 0 ] || exit $rc
S='import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ("value","ms_per_step","index_build_sec","batched_queries_per_sec")}, round(d["roofline"]["frac"],3), round(d["roofline_query"]["frac"],3), round(d["roofline_batch"]["frac"],3), d["roofline_build"]["achieved"])'
python bench.py --no-cpu-baseline --no-live-traffic --lambda-mode feature --metric cosine --kernel rational > gpurun_out/q_feat.log 2>gpurun_out/q_feat.err && tail -1 gpurun_out/q_feat.log | python -c "$S"
