# extra bench data points: README GI setting (start=64: empty march loop) and a C4-scale scene (3 M Gaussians, SH 3)
set -e
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --start 64 > gpurun_out/bench_start64.json 2> gpurun_out/bench_x.err || { tail -30 gpurun_out/bench_x.err; exit 1; }
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --gaussians 3000000 --sh-degree 3 > gpurun_out/bench_c4scale.json 2>> gpurun_out/bench_x.err || { tail -30 gpurun_out/bench_x.err; exit 1; }
python bench.py --steps 30 --warmup 5 > gpurun_out/bench_default.json 2>> gpurun_out/bench_x.err || { tail -30 gpurun_out/bench_x.err; exit 1; }
for f in start64 c4scale default; do python -c "
import json
d=json.loads(open('gpurun_out/bench_$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'], d['config'].get('R'), d.get('cpu_baseline'))"; done
