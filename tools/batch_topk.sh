#!/bin/bash
# batched rate and single-query reruns inside as_search_batch at a deep topk (the scorer's proofs are the first to feel a loose error term)
cd "$(dirname "$0")/.."
for tk in 15 100 400; do
  env "$@" timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --no-threaded --steps 50 --warmup 10 --topk $tk 2>/dev/null | grep "^{" | \
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'topk=$tk batched q/s %.0f' % d['batched_queries_per_sec'], 'single q/s %.0f' % d['value'], 'searches counted', d['fallback_rate']['searches'])"
done
