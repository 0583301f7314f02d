set -e
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
python bench.py --steps 10 --warmup 3 > gpurun_out/bench_v0.json 2> gpurun_out/bench_v0.err
python bench.py --steps 10 --warmup 3 --start 64 --no-cpu-baseline > gpurun_out/bench_v0_start64.json 2>> gpurun_out/bench_v0.err
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_v0 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_v0.log 2>&1
ls -R gpurun_out/prof_v0 | head -20
cat gpurun_out/smoke.log | tail -3
cat gpurun_out/bench_v0.json
