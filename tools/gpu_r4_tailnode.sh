# does a node on the launch stream after the backward graph's join shorten the graph exit when the side branch finishes last (C4)?
# (measured once, result in DESIGN.md section 9; the hook it switched -- GIGS_GB_TAIL_NODE=1: one gigs_stream_delay node behind
# pipeline.WholeStepGraph's backward capture -- was removed afterwards, so this script documents the experiment, it no longer runs it)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4t
mkdir -p $O
for rep in 1 2 3; do
for tn in 0 1; do
  GIGS_GB_TAIL_NODE=$tn python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --repeats 5 > $O/b.json 2>/dev/null
  python -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('c4 tail node $tn:', d['value'], d['repeats']['ms_per_step_median'], d['repeats']['ms_per_step_min'])"
done
done
for tn in 0 1; do
  GIGS_GB_TAIL_NODE=$tn python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-extras --repeats 5 > $O/b.json 2>/dev/null
  python -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('c2 tail node $tn:', d['value'], d['repeats']['ms_per_step_median'], d['repeats']['ms_per_step_min'])"
done
export GIGS_GB_TAIL_NODE=1
rocprofv3 --kernel-trace --output-format csv -d $O/prof_c4 -- python3 bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --repeats 1 > $O/prof_c4.log 2>&1 || { tail -20 $O/prof_c4.log; exit 1; }
f=$(find $O/prof_c4 -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f --all-queues --min-us 0 --step 5 > $O/c4_step_timeline_tailnode.txt
rm -rf $O/prof_c4
