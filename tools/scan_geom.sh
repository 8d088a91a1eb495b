#!/bin/bash
# single-query scan launch geometry A/B on one box: ARROWSPACE_SCAN_GEOM=<blocks per CU><ring slots> (rows of up to 512 image floats)
cd "$(dirname "$0")/.."
for g in ${GEOMS:-0 24 34 44}; do
  ARROWSPACE_SCAN_GEOM=$g timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --no-threaded --steps 200 --warmup 20 "$@" 2>/dev/null | grep '^{"metric"' | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('geom=$g', 'q/s=%.1f' % d['value'], 'scan ms=%.4f' % d['roofline']['avg_launch_ms'], 'moved frac=%.3f' % d['roofline']['frac_bytes_moved'])"
done
