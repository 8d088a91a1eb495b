set -e
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/bprof
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/bprof -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/batch_bench.py 1048576 768 256 > $GRAFT_REPO_ROOT/gpurun_out/bprof.log 2>&1
tail -2 $GRAFT_REPO_ROOT/gpurun_out/bprof.log
cd $GRAFT_REPO_ROOT && python profiles/summarise.py stats gpurun_out/bprof gpurun_out/bprof_stats.csv
