# rocprofv3 kernel-trace summary of the default bench command (hipGraphs on) and of the eager variant.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-v5}
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_${TAG}.log 2>&1 || { tail -20 gpurun_out/prof_${TAG}.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_eager -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --graphs off > gpurun_out/prof_${TAG}_eager.log 2>&1 || { tail -20 gpurun_out/prof_${TAG}_eager.log; exit 1; }
find gpurun_out/prof_${TAG} gpurun_out/prof_${TAG}_eager -name "*kernel_stats.csv"
tail -1 gpurun_out/prof_${TAG}.log
