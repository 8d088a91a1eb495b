#!/bin/bash
# gang geometry sweep of the bf16 K2 kernel: GC column pieces x 32/GC row blocks per XCD
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
N=${1:-1000000}
for gc in 4 8 16; do
  echo "GC=$gc"; ARROWSPACE_K2_GANG_GC=$gc timeout -k 10 300 python tools/build_only.py $N 1 2>&1 | tail -1 || exit 1
done
echo nogang; ARROWSPACE_K2_NO_GANG=1 timeout -k 10 300 python tools/build_only.py $N 1 2>&1 | tail -1
