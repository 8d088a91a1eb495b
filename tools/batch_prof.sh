#!/bin/bash
# Runs on the GPU box: kernel-trace stats of the batched passes (tools/batch_bench.py with the bench's k / topk) -- the batched scan and the kernels behind it.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
rm -rf gpurun_out/prof_batch
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_batch -- python3 bench.py --no-cpu-baseline --no-live-traffic --steps 20 --warmup 2 > gpurun_out/prof_batch.log 2>&1 || { tail -5 gpurun_out/prof_batch.log; exit 1; }
python3 profiles/summarise.py stats gpurun_out/prof_batch gpurun_out/batch_stats.csv | grep -E "kernel,|scan_gemm|gmin_batch|filter_batch|knn_finish|score_finish|pick_thr|q_prepare"
