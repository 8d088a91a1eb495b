set -e
mkdir -p gpurun_out
ARROWSPACE_BENCH_FORCE_DIST=1 timeout -k 10 400 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > gpurun_out/bench_dist1.log 2>&1 || { tail -20 gpurun_out/bench_dist1.log; exit 1; }
grep '^{"metric"' gpurun_out/bench_dist1.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','index_build_sec')}); print(d['roofline']['avg_launch_ms'])"
