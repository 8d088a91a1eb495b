#!/bin/bash
# host time of the two halves of a batched pass (ARROWSPACE_DEBUG stage line of as_search_batch) at the shapes of batch_wide.sh
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for shape in "65536 4096" "262144 1024" "1048576 768"; do
  set -- $shape
  ARROWSPACE_DEBUG=1 timeout -k 10 300 python tools/batch_bench.py $1 $2 256 2>&1 | grep -E "as_search_batch:|variant=" | tail -4
done
