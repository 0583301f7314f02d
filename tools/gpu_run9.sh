set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/test9.log 2>&1 || { tail -60 gpurun_out/test9.log; exit 1; }
tail -2 gpurun_out/test9.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_v9.json 2> gpurun_out/bench_v9.err || { tail -30 gpurun_out/bench_v9.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/bench_v9.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])"
