#!/bin/bash
# a fuzz campaign on the GPU box: plain, boundary sizes, feature mode, the staged (sharded) path -- fresh seeds
# bash tools/gpu_fuzz_campaign.sh <seed0> [cases]
set -o pipefail
S=${1:-300}; N=${2:-150}
mkdir -p gpurun_out
rc=0
timeout -k 10 280 python tools/fuzz_parity.py $N $S > gpurun_out/fuzz_plain.log 2>&1 || rc=1; tail -2 gpurun_out/fuzz_plain.log
FUZZ_BOUNDARIES=1 timeout -k 10 280 python tools/fuzz_parity.py $N $((S+1)) > gpurun_out/fuzz_bound.log 2>&1 || rc=1; tail -2 gpurun_out/fuzz_bound.log
FUZZ_FEATURE=1 timeout -k 10 280 python tools/fuzz_parity.py $N $((S+2)) > gpurun_out/fuzz_feat.log 2>&1 || rc=1; tail -2 gpurun_out/fuzz_feat.log
timeout -k 10 280 python tools/fuzz_parity.py $((N/2)) $((S+3)) - sharded > gpurun_out/fuzz_shard.log 2>&1 || rc=1; tail -2 gpurun_out/fuzz_shard.log
exit $rc
