# rocprofv3 kernel-trace summary of one full training iteration variant (tools/train_iter_bench.py --only ...)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
WHICH=${1:-stage1_hip}
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_iter_${WHICH} -- python3 $GRAFT_REPO_ROOT/tools/train_iter_bench.py --only ${WHICH} --steps 20 --warmup 5 > gpurun_out/prof_iter_${WHICH}.log 2>&1 || { tail -20 gpurun_out/prof_iter_${WHICH}.log; exit 1; }
find gpurun_out/prof_iter_${WHICH} -name "*kernel_stats.csv"
