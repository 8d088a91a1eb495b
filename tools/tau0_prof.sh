#!/bin/bash
# kernel-trace stats of single-query searches at a given tau: tools/tau0_prof.sh <tau> <tag>
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp && cd "$ROOT" || exit 1
TAU=${1:-0}; TAG=${2:-t0}
rm -rf gpurun_out/prof_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --no-cpu-baseline --no-live-traffic --tau $TAU --steps 100 --warmup 5 --no-threaded > gpurun_out/prof_$TAG.log 2>&1 || { tail -5 gpurun_out/prof_$TAG.log; exit 1; }
python3 profiles/summarise.py stats gpurun_out/prof_$TAG gpurun_out/${TAG}_stats.csv | grep -v -E "knn_bf16|refine|sym_|energy|transposed|split|degree|lambda_kernel|rowlen|sel_|ingest|scan_gemm|_batch|scan_block|scan_top|scan_apply|reset_info" | head -20
python3 -c "
import json;d=json.loads([l for l in open('gpurun_out/prof_$TAG.log') if l.startswith('{\"metric')][-1]);print('tau $TAU: q/s', round(d['value']), 'in-dist', round(d['in_distribution_queries']['value']), 'zero', d['zero_lambda_rate'])"
