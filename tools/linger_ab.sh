#!/bin/bash
# Runs on the GPU box: how long a gang's leader waits (from its turn) for the callers it expects -- Python threads arrive spread
# out by the interpreter lock, native threads do not.   bash tools/linger_ab.sh
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT" || exit 1
J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d["threaded_queries_per_sec"]; print(sys.argv[1], "B=1 %.0f" % d["value"], "py2 %.0f py4 %.0f n2 %.0f n4 %.0f" % (t["2"]["value"], t["4"]["value"], t["native_2"]["value"], t["native_4"]["value"]), t["gang_scans_by_members"])'
for L in 60 120 200 60 120; do
  ARROWSPACE_GANG_LINGER_US=$L python bench.py --no-cpu-baseline --no-distributions --no-host-build --no-live-traffic --verify-queries 0 2>/dev/null | python -c "$J" "linger $L us"
done
