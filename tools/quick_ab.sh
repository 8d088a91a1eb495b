#!/bin/bash
# three shapes, single-query rate only (no CPU baseline, no traffic passes, no threads): bash tools/quick_ab.sh [env assignments...]
cd "$(dirname "$0")/.."
for shape in "--n 1000000 --d 768" "--n 200000 --d 768" "--n 400000 --d 384 --k 4 --topk 2"; do
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --no-threaded --steps 300 --warmup 30 $shape 2>/dev/null | grep '^{"metric"' | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '$shape', 'q/s=%.1f' % d['value'], 'in-dist q/s=%.1f' % d['in_distribution_queries']['value'], 'batched=%.0f' % d['batched_queries_per_sec'], 'reruns', d['fallback_rate']['searches_with_rerun'])"
done
