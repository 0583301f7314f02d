set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/test3.log 2>&1 || { tail -60 gpurun_out/test3.log; exit 1; }
tail -3 gpurun_out/test3.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_v2.json 2> gpurun_out/bench_v2.err || { tail -30 gpurun_out/bench_v2.err; exit 1; }
bash tools/gpu_pmc.sh
