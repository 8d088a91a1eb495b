# measurement: the staged (N > 1) single-query path on one rank -- library-side RCCL exchange vs torch.distributed collectives -- next to the fused path
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"],1), "q/s", round(d["ms_per_step"]*1e3,1), "us/query; scan", round(d["roofline"]["avg_launch_ms"]*1e3,1))'
ARROWSPACE_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-live-traffic --steps 300 "$@" 2>/dev/null | python -c "$J" "staged, library RCCL     "
ARROWSPACE_PY_COLLECTIVES=1 ARROWSPACE_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-live-traffic --steps 300 "$@" 2>/dev/null | python -c "$J" "staged, torch collectives"
python bench.py --no-cpu-baseline --no-live-traffic --steps 300 "$@" 2>/dev/null | python -c "$J" "fused (one GPU)          "
