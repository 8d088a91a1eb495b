# measurement: the single-query scan kernel, plain against the SC variant (ARROWSPACE_SC_DBG=7: its skeleton alone), interleaved
# repetitions on one box (the same kernel varies by +-8 us from run to run)
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"],1), "q/s  scan", round(d["roofline"]["avg_launch_ms"]*1e3,1), "us")'
for rep in 1 2 3; do
  ARROWSPACE_NO_FUSED_TAIL=1 python bench.py --no-cpu-baseline --no-live-traffic --steps 300 "$@" 2>/dev/null | python -c "$J" "plain   "
  ARROWSPACE_SC_DBG=7 python bench.py --no-cpu-baseline --no-live-traffic --steps 300 "$@" 2>/dev/null | python -c "$J" "SC dbg 7"
  python bench.py --no-cpu-baseline --no-live-traffic --steps 300 "$@" 2>/dev/null | python -c "$J" "SC      "
done
