# rocprofv3 kernel-trace summary of the training-loop glue micro-benchmark (tools/glue_bench.py)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-v19}
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_glue_${TAG} -- python3 $GRAFT_REPO_ROOT/tools/glue_bench.py --iters 30 > gpurun_out/prof_glue_${TAG}.log 2>&1 || { tail -20 gpurun_out/prof_glue_${TAG}.log; exit 1; }
find gpurun_out/prof_glue_${TAG} -name "*kernel_stats.csv"
tail -1 gpurun_out/prof_glue_${TAG}.log
