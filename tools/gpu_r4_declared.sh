set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4m
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_training.py tests/test_gpu_pbr.py tests/test_cabi.py -x -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for mo in 0 1; do
  GIGS_MATERIALS_ONLY=$mo timeout -k 10 300 python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_mo$mo.json 2> $O/bench_c4_mo$mo.err || { tail -30 $O/bench_c4_mo$mo.err; exit 1; }
  GIGS_MATERIALS_ONLY=$mo timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_c2_mo$mo.json 2> $O/bench_c2_mo$mo.err || { tail -30 $O/bench_c2_mo$mo.err; exit 1; }
  for c in c2 c4; do python -c "
import json
d=json.loads(open('$O/bench_${c}_mo$mo.json').read().strip().splitlines()[-1])
print('$c materials_only=$mo: step', d['value'], 'iteration', d['iteration']['iterations_per_s'], d['iteration']['ms_per_iteration'], 'cached', d['iteration_cached_geometry']['iterations_per_s'], 'stage1', d['iteration_stage1']['iterations_per_s'])"; done
done
