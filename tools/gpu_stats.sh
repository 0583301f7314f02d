set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_v3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_v3.log 2>&1 || { tail -20 gpurun_out/prof_v3.log; exit 1; }
ls gpurun_out/prof_v3/*/
