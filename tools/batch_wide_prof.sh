#!/bin/bash
# kernel-trace stats of the batched passes at N x D (tools/batch_bench.py): usage batch_wide_prof.sh N D tag
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
N=$1; D=$2; TAG=${3:-w}
rm -rf gpurun_out/prof_bw_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_bw_$TAG -- python3 tools/batch_bench.py $N $D 256 > gpurun_out/prof_bw_$TAG.log 2>&1 || { tail -5 gpurun_out/prof_bw_$TAG.log; exit 1; }
python3 profiles/summarise.py stats gpurun_out/prof_bw_$TAG gpurun_out/bw_${TAG}_stats.csv | grep -E "kernel,|scan_gemm|gemm_combine|gmin_batch|filter_batch|knn_finish|score_finish|pick_thr|q_prepare"
