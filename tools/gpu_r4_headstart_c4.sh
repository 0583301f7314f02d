set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4hs
mkdir -p $O
for rep in 1 2; do
for hs in 10 0 40 150; do
  GIGS_LIGHT_BWD_HEAD_START_US=$hs timeout -k 10 300 python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/b.json 2> $O/b.err || { tail -30 $O/b.err; exit 1; }
  python -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('c4 head start $hs us: step', d['value'], d['repeats']['ms_per_step_median'], d['repeats']['ms_per_step_min'])"
done
done
