set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/test2.log 2>&1 || { tail -60 gpurun_out/test2.log; exit 1; }
tail -5 gpurun_out/test2.log
python bench.py --steps 10 --warmup 3 > gpurun_out/bench_v1.json 2> gpurun_out/bench_v1.err || { tail -30 gpurun_out/bench_v1.err; exit 1; }
cat gpurun_out/bench_v1.json
