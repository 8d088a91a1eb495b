#!/bin/bash
# Runs on the GPU box: batched search, round 3's pass (bf16 head + tail products on the bf16 matrix pipe, fp16 cosines in the
# dots' places) against the fp32 pass (ARROWSPACE_BATCH_F32_DOTS=1), alternating on one box.
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT" || exit 1
for i in 1 2; do
  echo -n "fp32 products, fp32 dots : "; ARROWSPACE_BATCH_F32_DOTS=1 python tools/batch_bench.py "${1:-1000000}" "${2:-768}" "${3:-1024}" 2>&1 | grep "q/s"
  echo -n "bf16x3 products, fp16 cos: "; python tools/batch_bench.py "${1:-1000000}" "${2:-768}" "${3:-1024}" 2>&1 | grep "q/s"
done
