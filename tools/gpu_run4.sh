set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/test4.log 2>&1 || { tail -60 gpurun_out/test4.log; exit 1; }
tail -3 gpurun_out/test4.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_v3.json 2> gpurun_out/bench_v3.err || { tail -30 gpurun_out/bench_v3.err; exit 1; }
