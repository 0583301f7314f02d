set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_pbr.py -m gpu -x -q 2>&1 | tail -2
for b in 100000 512 256 128; do
  GIGS_SHADE_BWD_BLOCKS=$b python bench.py --steps 10 --warmup 3 --no-cpu-baseline --graphs off > gpurun_out/bench_sb$b.json 2> gpurun_out/bench_sb.err || { tail -30 gpurun_out/bench_sb.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/bench_sb$b.json').read().strip().splitlines()[-1])
print($b, d['kernels']['shade_bwd']['ms_per_step'], d['kernels']['shade_fwd']['ms_per_step'])"
done
