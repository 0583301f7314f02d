set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_pbr.py -m gpu -x -q > gpurun_out/test6.log 2>&1 || { tail -80 gpurun_out/test6.log; exit 1; }
tail -2 gpurun_out/test6.log
for ab in 0 1 2 3; do
  GIGS_ABLATE=$ab python bench.py --steps 10 --warmup 3 --no-cpu-baseline --graphs off > gpurun_out/bench_ab$ab.json 2> gpurun_out/bench_ab.err || { tail -30 gpurun_out/bench_ab.err; exit 1; }
done
