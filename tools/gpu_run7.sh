set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_pbr.py -m gpu -x -q 2>&1 | tail -2
for m in 0 128 600; do
  GIGS_SPEC_MAX8=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline --graphs off > gpurun_out/bench_m$m.json 2> gpurun_out/bench_m.err || { tail -30 gpurun_out/bench_m.err; exit 1; }
  python -c "
import json,sys
d=json.loads(open('gpurun_out/bench_m$m.json').read().strip().splitlines()[-1])
print($m, d['kernels']['cubemap_fwd']['ms_per_step'], d['kernels']['cubemap_bwd']['ms_per_step'])"
done
