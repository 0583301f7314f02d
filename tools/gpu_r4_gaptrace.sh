# what runs between the end of the backward graph and the next step's first kernel at C4?  kernel + memory-copy + HIP API traces
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4t
mkdir -p $O
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $O/prof_gap -- python3 bench.py --config c4 --steps 6 --warmup 3 --no-cpu-baseline --no-extras --repeats 1 > $O/prof_gap.log 2>&1 || { tail -20 $O/prof_gap.log; exit 1; }
find $O/prof_gap -name "*.csv" | xargs ls -la
echo traced
