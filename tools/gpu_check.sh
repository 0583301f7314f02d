set -e
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -60 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_check.json 2> gpurun_out/bench_check.err || { tail -30 gpurun_out/bench_check.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/bench_check.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'])"
