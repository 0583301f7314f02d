# round-end artefacts: GPU tests, rocprofv3 stats (graphs on / eager), PMC passes, default bench line (tag = $1)
set -e
cd $GRAFT_REPO_ROOT
TAG=${1:-vX}
bash tools/gpu_check.sh
rm -rf gpurun_out/pmc
bash tools/gpu_stats.sh $TAG
bash tools/gpu_pmc.sh > gpurun_out/pmc_run.log 2>&1 || { tail -20 gpurun_out/pmc_run.log; exit 1; }
python bench.py --steps 30 --warmup 5 > gpurun_out/bench_${TAG}_default.json 2> gpurun_out/bench_${TAG}.err
python -c "
import json
d=json.loads(open('gpurun_out/bench_${TAG}_default.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['cpu_baseline']['value'])
print({k:round(v['ms_per_step'],3) for k,v in d['kernels'].items()})"
