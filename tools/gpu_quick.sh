#!/bin/bash
# parity subset + three short benches (default, tau=0, 400k x 384)
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_edges.py tests/test_gpu_golden.py tests/test_gpu_dist.py tests/test_gpu_fuzz.py tests/test_gpu_feature.py -x -q --timeout 200 --timeout-method=thread > gpurun_out/r2_tests_b.log 2>&1; rc=$?
tail -4 gpurun_out/r2_tests_b.log
[ $rc -eq 0 ] || exit $rc
S='import sys,json; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ("value","ms_per_step","index_build_sec","batched_queries_per_sec")}, round(d["roofline"]["frac"],3), round(d["roofline_query"]["frac"],3), round(d["roofline_batch"]["frac"],3))'
python bench.py --no-cpu-baseline --no-live-traffic > gpurun_out/q_default.log 2>gpurun_out/q_default.err && tail -1 gpurun_out/q_default.log | python -c "$S" &&
python bench.py --no-cpu-baseline --no-live-traffic --tau 0 > gpurun_out/q_tau0.log 2>gpurun_out/q_tau0.err && tail -1 gpurun_out/q_tau0.log | python -c "$S" &&
python bench.py --no-cpu-baseline --no-live-traffic --n 400000 --d 384 --k 4 --topk 2 > gpurun_out/q_400k.log 2>gpurun_out/q_400k.err && tail -1 gpurun_out/q_400k.log | python -c "$S" &&
python bench.py --no-cpu-baseline --no-live-traffic --n 200000 --d 768 > gpurun_out/q_200k.log 2>gpurun_out/q_200k.err && tail -1 gpurun_out/q_200k.log | python -c "$S"
