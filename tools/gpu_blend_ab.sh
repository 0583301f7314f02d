# eager bench, repeated: blend kernel times are noisy box to box, compare within one box
set -e
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline --graphs off > gpurun_out/bench_bl.json 2> gpurun_out/bench_bl.err || { tail -20 gpurun_out/bench_bl.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/bench_bl.json').read().strip().splitlines()[-1])
print('blend_fwd', d['kernels']['blend_fwd']['ms_per_step'], 'blend_bwd', d['kernels']['blend_bwd']['ms_per_step'])"
done
