# A/B of the hit-byte layout ([R][4] -> [4][R]) on one box: parity tests, then C4 and C2 with the new and the previous library
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4j
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -k "backward or forward or stage2 or c4 or c2_full or ragged" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for rep in 1 2; do
for lib in "" gpurun_in/lib_old.so; do
  if [ -n "$lib" ]; then export GIGS_LIB=$GRAFT_REPO_ROOT/$lib; else unset GIGS_LIB; fi
  for cfg in c4 c2; do
  python bench.py --config $cfg --steps 20 --warmup 5 --no-cpu-baseline --no-extras --repeats 3 > $O/bench_cmp.json 2>/dev/null
  python -c "
import json
d=json.loads(open('$O/bench_cmp.json').read().strip().splitlines()[-1])
print('$cfg lib=${lib:-new}', d['value'], d['repeats']['ms_per_step_median'], {k:round(v['avg_ms'],4) for k,v in d['kernels'].items() if 'blend' in k})"
  done
done
done
