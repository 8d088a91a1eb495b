#!/bin/bash
# one GPU-box session: the -m gpu suite (streamed to a file so a hang shows where), then optional extras
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -v --timeout 200 --timeout-method=thread --durations=12 > gpurun_out/r2_tests.log 2>&1; rc=$?
tail -30 gpurun_out/r2_tests.log
exit $rc
