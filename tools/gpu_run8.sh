set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/test8.log 2>&1 || { tail -60 gpurun_out/test8.log; exit 1; }
tail -2 gpurun_out/test8.log
for cfg in "on on" "on off" "off on"; do
  set -- $cfg
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --fused $1 --graphs $2 > gpurun_out/bench_f$1_g$2.json 2> gpurun_out/bench_f.err || { tail -30 gpurun_out/bench_f.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/bench_f$1_g$2.json').read().strip().splitlines()[-1])
print('fused=$1 graphs=$2', d['value'], d['ms_per_step'])"
done
