# measurement: cost of the parts of the scan-side scorer candidates (ARROWSPACE_SC_DBG: 1 no publication, 2 no histogram read, 4 no candidates)
# and the plain chain on the same box (ARROWSPACE_NO_FUSED_TAIL)
for D in ${SC_PROBE_SET:-0 7}; do
  ARROWSPACE_SC_DBG=$D python bench.py --no-cpu-baseline --no-live-traffic --steps 100 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dbg', $D, d['value'], d['roofline']['avg_launch_ms'], d['in_distribution_queries']['value'], d['roofline_query']['frac'])"
done
ARROWSPACE_NO_FUSED_TAIL=1 python bench.py --no-cpu-baseline --no-live-traffic --steps 100 --warmup 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('plain', d['value'], d['roofline']['avg_launch_ms'], d['in_distribution_queries']['value'], d['roofline_query']['frac'])"
