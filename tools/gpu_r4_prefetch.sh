set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4p
mkdir -p $O
python -m pytest tests/test_gpu_pbr.py tests/test_gpu_training.py tests/test_gpu_relight.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for rep in 1 2; do
for pf in 0 1; do
  GIGS_LIGHT_PREFETCH=$pf python bench.py --fused off --graphs off --no-cpu-baseline --no-extras --steps 40 --warmup 10 > $O/b.json 2>/dev/null
  python -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('drop-in, prefetch $pf:', d['value'], d['repeats']['ms_per_step_median'], d['repeats']['ms_per_step_min'])"
done
done
rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --fused off --graphs off --steps 12 --warmup 4 --no-cpu-baseline --no-extras --repeats 1 > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f --all-queues --min-us 0 --step 8 > $O/dropin_step_timeline_prefetch.txt
rm -rf $O/prof
