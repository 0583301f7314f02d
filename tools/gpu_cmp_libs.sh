set -e
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for lib in "" gpurun_in/lib_old.so; do
  if [ -n "$lib" ]; then export GIGS_LIB=$GRAFT_REPO_ROOT/$lib; else unset GIGS_LIB; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_cmp.json 2>/dev/null
  python -c "
import json
d=json.loads(open('gpurun_out/bench_cmp.json').read().strip().splitlines()[-1])
print('lib=$lib', d['value'], {k:round(v['ms_per_step'],3) for k,v in d['kernels'].items() if 'blend' in k})"
done
done
