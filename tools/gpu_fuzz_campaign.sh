#!/bin/bash
# a fuzz campaign on the GPU box, fresh seeds: bash tools/gpu_fuzz_campaign.sh <seed0> [cases]
# fused path (plain, boundary sizes, feature mode, entry points), staged path (plain, real RCCL on one rank), ring build
# over random blocks, larger sizes, 2 / 3 / 4 ranks on one GPU
set -o pipefail
S=${1:-300}; N=${2:-150}
mkdir -p gpurun_out
rc=0
run() { # name, timeout, command...
    local name=$1 t=$2; shift 2
    timeout -k 10 $t "$@" > gpurun_out/fuzz_$name.log 2>&1 || rc=1
    grep -h "FAIL" gpurun_out/fuzz_$name.log | cut -c1-600 | head -5
    tail -1 gpurun_out/fuzz_$name.log | cut -c1-300
}
run plain 280 python tools/fuzz_parity.py $N $S
FUZZ_BOUNDARIES=1 run bound 280 python tools/fuzz_parity.py $N $((S+1))
FUZZ_FEATURE=1 FUZZ_EXTRAS=1 run feat 280 python tools/fuzz_parity.py $N $((S+2))
run shard 280 python tools/fuzz_parity.py $N $((S+3)) - sharded
FUZZ_RCCL=1 FUZZ_FEATURE=1 run rccl 280 python tools/fuzz_parity.py $((N/2)) $((S+4)) - sharded
run ring 280 python tools/fuzz_ring.py $((N/2)) $((S+5))
run big_build 280 python tools/fuzz_big.py $((N/20)) $((S+6)) build
run big_search 280 python tools/fuzz_big.py $((N/10)) $((S+7)) search
FUZZ_FEATURE=1 run 2rank 280 python tools/fuzz_2rank.py $((N/4)) $((S+8)) 2
run 3rank 280 python tools/fuzz_2rank.py $((N/10)) $((S+9)) 3
run 4rank 280 python tools/fuzz_2rank.py $((N/10)) $((S+10)) 4
exit $rc
