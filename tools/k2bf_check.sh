#!/bin/bash
# bf16 head + tail K2: parity subset on the default (bf16) path, then build times: gang order, no gang order, fp32 pipe
set -o pipefail
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_symmetric.py tests/test_gpu_edges.py tests/test_gpu_ring.py -x -q --timeout 300 --timeout-method=thread > gpurun_out/k2bf_tests.log 2>&1; rc=$?
tail -5 gpurun_out/k2bf_tests.log
[ $rc -eq 0 ] || exit $rc
N1=${N1:-262144}
timeout -k 10 300 python tools/build_only.py $N1 2 > gpurun_out/k2bf_build_256k.log 2>&1 && tail -2 gpurun_out/k2bf_build_256k.log &&
ARROWSPACE_K2_NO_GANG=1 timeout -k 10 300 python tools/build_only.py $N1 2 > gpurun_out/k2bf_nogang_256k.log 2>&1 && tail -1 gpurun_out/k2bf_nogang_256k.log &&
timeout -k 10 400 python tools/build_only.py 1000000 2 > gpurun_out/k2bf_build_1m.log 2>&1 && tail -2 gpurun_out/k2bf_build_1m.log &&
ARROWSPACE_K2_NO_GANG=1 timeout -k 10 400 python tools/build_only.py 1000000 1 > gpurun_out/k2bf_nogang_1m.log 2>&1 && tail -1 gpurun_out/k2bf_nogang_1m.log
