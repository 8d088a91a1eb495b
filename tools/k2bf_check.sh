#!/bin/bash
# bf16 head + tail K2: parity subset on the default (bf16) path, then build times bf16 vs fp32 (ARROWSPACE_K2_FP32=1)
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_symmetric.py tests/test_gpu_edges.py tests/test_gpu_ring.py -x -q --timeout 300 --timeout-method=thread > gpurun_out/k2bf_tests.log 2>&1; rc=$?
tail -5 gpurun_out/k2bf_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/build_only.py 262144 2 > gpurun_out/k2bf_build_256k.log 2>&1 && tail -2 gpurun_out/k2bf_build_256k.log &&
ARROWSPACE_K2_FP32=1 timeout -k 10 300 python tools/build_only.py 262144 1 > gpurun_out/k2f32_build_256k.log 2>&1 && tail -1 gpurun_out/k2f32_build_256k.log &&
timeout -k 10 400 python tools/build_only.py 1000000 2 > gpurun_out/k2bf_build_1m.log 2>&1 && tail -2 gpurun_out/k2bf_build_1m.log
