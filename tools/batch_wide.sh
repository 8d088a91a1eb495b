#!/bin/bash
# batched search on rows wider than 768 floats (K-chunk passes): parity, then throughput at the shapes of profiles/r03_sweep.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "batch" --timeout 300 --timeout-method=thread 2>&1 | tail -3 || exit 1
for shape in "262144 1024" "131072 2048" "65536 4096" "131072 1536" "1048576 768"; do
  set -- $shape
  timeout -k 10 300 python tools/batch_bench.py $1 $2 256 2>&1 | tail -1
done
