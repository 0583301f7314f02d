set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4sp
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_contract.py tests/test_gpu_training.py tests/test_gpu_pbr.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for sp in 0 1; do
  for c in c4 c2; do
  GIGS_SPLIT_SH=$sp timeout -k 10 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline > $O/b.json 2> $O/b.err || { tail -30 $O/b.err; exit 1; }
  python -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$c split_sh=$sp: step', d['value'], 'iteration', d['iteration']['iterations_per_s'], d['iteration']['ms_per_iteration'], 'cached', d['iteration_cached_geometry']['iterations_per_s'], 'final loss', d['iteration']['final_loss'])"
  done
done
