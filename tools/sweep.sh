#!/bin/bash
# Headline-adjacent configurations (SURVEY 8d "config restatement" + width/topk/tau variations): one bench.py line each.
set -e
mkdir -p gpurun_out
out=gpurun_out/sweep.jsonl
: > $out
run() { timeout -k 10 400 python bench.py --no-cpu-baseline --no-live-traffic --no-distributions --no-host-build --steps 100 --warmup 10 "$@" 2>/dev/null | grep '^{"metric"' >> $out; }
run --n 400000 --d 384 --k 4 --topk 2
run --n 200000 --d 768
run --n 1000000 --d 768 --topk 100
run --n 1000000 --d 768 --tau 1.0
run --n 1000000 --d 768 --tau 0.0
run --n 262144 --d 1024
run --n 131072 --d 2048
run --n 65536 --d 4096
run --n 20000 --d 768
python - <<'PY'
import json, re
for l in open("gpurun_out/sweep.jsonl"):
    d = json.loads(l)
    c = d["config"]
    m = re.search(r"k=(\d+) topk=(\d+) tau=([0-9.]+)", c["workload"])
    print(f"n={c['n']:>8} d={c['d']:>5} k={m.group(1):>3} topk={m.group(2):>4} tau={m.group(3)} q/s={d['value']:9.1f} scan={d['roofline']['frac']:.3f} query={d['roofline_query']['frac']:.3f} "
          f"in-dist q/s={d['in_distribution_queries']['value']:9.1f} build={d['index_build_sec']:.2f}s mfma={d['roofline_build']['frac']:.3f} batched={d['batched_queries_per_sec'] or 0:.0f} "
          f"(batch frac {d['roofline_batch']['frac'] if d['roofline_batch'] else 0:.3f}) "
          f"threads 2/4: {d['threaded_queries_per_sec']['2']['value']:.0f} ({d['threaded_queries_per_sec']['2']['frac']:.3f}) / "
          f"{d['threaded_queries_per_sec']['4']['value']:.0f} ({d['threaded_queries_per_sec']['4']['frac']:.3f}) "
          f"native 4: {d['threaded_queries_per_sec'].get('native_4', {}).get('value', 0):.0f} zero-lambda {d['zero_lambda_rate']:.2f} "
          f"reruns {d['fallback_rate']['rate']:.3f} verified {d['verified']['n']}/{d['verified']['mismatches']} bad operand {d['roofline']['operand'][:22]}")
PY
