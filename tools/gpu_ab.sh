#!/bin/bash
# same-box A/B of two builds of the library: pyarrowspace_amd/libarrowspace_hip.so (new) against
# pyarrowspace_amd/libarrowspace_hip_old.so (built by hand from an earlier source), alternating runs
set -o pipefail
mkdir -p gpurun_out
L=pyarrowspace_amd/libarrowspace_hip.so
cp $L gpurun_out/lib_new.so
S='import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["ms_per_step"]*1000,1), d["batched_queries_per_sec"] and round(d["batched_queries_per_sec"]))'
for ARGS in "--n 1000000 --d 768" "--n 400000 --d 384 --k 4 --topk 2" "--n 200000 --d 768"; do
  for rep in 1 2; do
    for which in new old; do
      if [ $which = new ]; then cp gpurun_out/lib_new.so $L; else cp pyarrowspace_amd/libarrowspace_hip_old.so $L; fi
      echo -n "$ARGS $which: "
      timeout -k 10 300 python bench.py --no-cpu-baseline --no-live-traffic --steps 400 $ARGS 2>gpurun_out/ab.err | tail -1 | python -c "$S" || { cp gpurun_out/lib_new.so $L; exit 1; }
    done
  done
done
cp gpurun_out/lib_new.so $L
