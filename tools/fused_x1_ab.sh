#!/bin/bash
# the fused tail as one 1024-thread block (default) against its two-launch form (ARROWSPACE_FUSED_X1=1), same box, interleaved
cd "$(dirname "$0")/.."
for rep in 1 2; do
for v in 0 1; do
  for shape in "--n 1000000 --d 768" "--n 200000 --d 768" "--n 400000 --d 384 --k 4 --topk 2"; do
    ARROWSPACE_FUSED_X1=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --no-threaded --steps 300 --warmup 30 $shape 2>/dev/null | grep '^{"metric"' | \
      python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('fused_x1=$v', '$shape', 'q/s=%.1f' % d['value'], 'in-dist q/s=%.1f' % d['in_distribution_queries']['value'], 'reruns', d['fallback_rate']['searches_with_rerun'])"
  done
done
done
