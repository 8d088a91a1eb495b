set -e
mkdir -p gpurun_out
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -40 gpurun_out/gpu_tests.log | cut -c1-250; exit 1; }
tail -3 gpurun_out/gpu_tests.log
bash tools/sweep.sh
