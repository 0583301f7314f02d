# renders/s of the whole-step graph as a function of the light backward's head-start delay
cd $GRAFT_REPO_ROOT
for us in 0 3 6 10 20; do
  GIGS_LIGHT_BWD_HEAD_START_US=$us python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('head start $us us:', d['value'], d['ms_per_step'])"
done
GIGS_STEP_GRAPH=0 python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('eager rasterizer:', d['value'], d['ms_per_step'])"
