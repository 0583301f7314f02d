# A/B of one environment switch inside one box: usage gpu_ab.sh VAR A B [steps]
set -e
cd $GRAFT_REPO_ROOT
VAR=$1; A=$2; B=$3; STEPS=${4:-100}
for rep in 1 2 3; do
  for v in $A $B; do
    env $VAR=$v python bench.py --steps $STEPS --warmup 5 --no-cpu-baseline > gpurun_out/bench_ab.json 2> gpurun_out/bench_ab.err || { tail -30 gpurun_out/bench_ab.err; exit 1; }
    python -c "
import json
d=json.loads(open('gpurun_out/bench_ab.json').read().strip().splitlines()[-1])
print('$VAR=$v', d['value'], d['ms_per_step'])"
  done
done
