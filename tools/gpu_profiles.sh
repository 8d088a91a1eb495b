#!/bin/bash
# round-2 profiles: kernel stats + PMC traffic for the headline workload in the three modes, the single-query
# timeline, and the 1-GPU rehearsal of the N>1 bench path.  Summaries land in gpurun_out/ (copied to profiles/ by hand).
set -o pipefail
bash profiles/collect.sh r02 || exit 1
bash profiles/collect.sh r02_cosine --metric cosine --kernel rational || exit 1
bash profiles/collect.sh r02_feature --lambda-mode feature --metric cosine --kernel rational || exit 1
bash tools/gpu_timeline.sh r02 || exit 1
ARROWSPACE_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-live-traffic --steps 100 > gpurun_out/r02_bench_force_dist.json 2> gpurun_out/r02_bench_force_dist.err || { tail -5 gpurun_out/r02_bench_force_dist.err; exit 1; }
tail -c 600 gpurun_out/r02_bench_force_dist.json
