#!/bin/bash
# kernel-trace timeline of the single-query chain: bash tools/gpu_timeline.sh <tag> [bench args]
set -o pipefail
TAG=$1; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"   # the repo this script lives in (never an unset variable: `cd ""` stays put)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
OUT=gpurun_out/tl_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --no-cpu-baseline --no-live-traffic --no-threaded --steps 200 --warmup 20 "$@" > $OUT/bench.log 2>&1 || { tail -5 $OUT/bench.log; exit 1; }
python3 profiles/summarise.py timeline $OUT/trace gpurun_out/${TAG}_timeline.csv
python3 profiles/summarise.py stats $OUT/trace gpurun_out/${TAG}_kernel_stats.csv > /dev/null
grep '^{"metric"' $OUT/bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','index_build_sec','batched_queries_per_sec')}, d['roofline']['frac'], d['roofline_query']['frac'], d['roofline_batch']['frac'])"
rm -rf $OUT/trace
