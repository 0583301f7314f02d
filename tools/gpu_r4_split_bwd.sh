# the fused node's backward as two launches (material gradients first, the light scatter on the light's stream): re-measured at C4,
# where the blend branch of the tail is the longer one
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4sb
mkdir -p $O
for rep in 1 2; do
for sp in 0 1; do
  for c in c4 c2; do
  GIGS_SHADE_BWD_SPLIT=$sp timeout -k 10 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/b.json 2> $O/b.err || { tail -30 $O/b.err; exit 1; }
  python -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$c split=$sp: step', d['value'], d['repeats']['ms_per_step_median'], d['repeats']['ms_per_step_min'])"
  done
done
done
