# timeline of the drop-in step (what an unmodified train.py gets: --fused off --graphs off) at C2: where is the GPU idle?
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4d
mkdir -p $O
true
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --fused off --graphs off --steps 12 --warmup 4 --no-cpu-baseline --no-extras --repeats 1 > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
f=$(find $O/prof -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f --all-queues --min-us 0 --step 8 > $O/dropin_step_timeline.txt
cp $(find $O/prof -name "*kernel_stats.csv" | head -1) $O/dropin_kernel_stats.csv
rm -rf $O/prof
exit 0
python -c "
import json
d=json.loads(open('$O/bench_dropin.json').read().strip().splitlines()[-1])
print('drop-in:', d['value'], d['ms_per_step'], d['repeats'])"
