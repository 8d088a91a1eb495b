J='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d["threaded_queries_per_sec"]; print(sys.argv[1], "B=1 %.0f" % d["value"], "batched %.0f" % d["batched_queries_per_sec"], "py2 %.0f py4 %.0f n4 %.0f" % (t["2"]["value"], t["4"]["value"], t["native_4"]["value"]))'
for i in 1 2; do
  python bench.py --no-cpu-baseline --no-distributions --no-host-build --no-live-traffic --verify-queries 0 2>/dev/null | python -c "$J" "default queues"
  GPU_MAX_HW_QUEUES=8 python bench.py --no-cpu-baseline --no-distributions --no-host-build --no-live-traffic --verify-queries 0 2>/dev/null | python -c "$J" "8 hw queues   "
done
