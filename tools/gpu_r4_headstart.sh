# does the light's backward still need its delay node now that the diffuse filter's backward (12 us) runs in front of it?
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4o
mkdir -p $O
for rep in 1 2 3; do
for hs in 10 0 5; do
  GIGS_LIGHT_BWD_HEAD_START_US=$hs python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras --repeats 5 > $O/b.json 2>/dev/null
  python -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('head start $hs us:', d['value'], d['repeats']['ms_per_step_median'], d['repeats']['ms_per_step_min'])"
done
done
