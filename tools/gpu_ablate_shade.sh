# shade backward: time with different LDS budgets (which light-gradient levels accumulate in LDS) and block counts
set -e
cd $GRAFT_REPO_ROOT
for cfg in "30720 256" "30720 1024" "9300 256" "9300 1024" "4700 1024" "0 1024"; do
  set -- $cfg
  GIGS_SHADE_LDS_FLOATS=$1 GIGS_SHADE_BWD_BLOCKS=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --graphs off > gpurun_out/bench_abl.json 2> gpurun_out/bench_abl.err || { tail -20 gpurun_out/bench_abl.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/bench_abl.json').read().strip().splitlines()[-1])
print('lds_floats=$1 blocks=$2 shade_bwd', d['kernels']['shade_bwd']['ms_per_step'])"
done
