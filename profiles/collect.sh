#!/bin/bash
# Runs on the GPU box (gpurun): kernel-trace stats of the default bench command, then the
# HBM traffic counters in separate --pmc passes (MI355X_MICROARCH.md, rocprofv3 section).
# Usage: bash profiles/collect.sh <round-tag> [bench args...]
set -o pipefail
TAG=${1:-r01}; shift
ROOT="$(cd "$(dirname "$0")/.." && pwd)"   # the repo this script lives in (never an unset variable: `cd ""` stays put)
cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT profiles
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --no-cpu-baseline --no-live-traffic --no-threaded --no-distributions --no-host-build --verify-queries 0 "$@" > $OUT/stats.log 2>&1 || exit 1
python3 profiles/summarise.py stats $OUT/stats gpurun_out/${TAG}_kernel_stats.csv
grep "^{\"metric\"" $OUT/stats.log | tail -1 > gpurun_out/${TAG}_bench_under_rocprof.json
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -- python3 bench.py --no-cpu-baseline --no-live-traffic --no-threaded --no-distributions --no-host-build --verify-queries 0 --steps 20 --warmup 2 "$@" > $OUT/pmc_$C.log 2>&1 || exit 1
  python3 profiles/summarise.py pmc $OUT/pmc_$C $C gpurun_out/${TAG}_pmc_$C.csv
done
python3 profiles/summarise.py traffic gpurun_out/${TAG}_pmc_FETCH_SIZE.csv gpurun_out/${TAG}_pmc_WRITE_SIZE.csv ${AS_PROFILE_N:-1000000} ${AS_PROFILE_D:-768} gpurun_out/${TAG}_traffic.json
