set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/test5.log 2>&1 || { tail -80 gpurun_out/test5.log; exit 1; }
tail -3 gpurun_out/test5.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graphs off > gpurun_out/bench_v4_eager.json 2> gpurun_out/bench_v4.err || { tail -30 gpurun_out/bench_v4.err; exit 1; }
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graphs on > gpurun_out/bench_v4_graph.json 2>> gpurun_out/bench_v4.err || { tail -30 gpurun_out/bench_v4.err; exit 1; }
