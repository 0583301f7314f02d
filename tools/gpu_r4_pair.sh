set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4pair
mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_contract.py -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for c in c4 c2 c4 c2; do
  timeout -k 10 300 python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/b.json 2> $O/b.err || { tail -30 $O/b.err; exit 1; }
  python -c "
import json
d=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
k=d['kernels']
print('$c: step', d['value'], d['repeats']['ms_per_step_median'], d['repeats']['ms_per_step_min'], 'blend_bwd', k['blend_bwd']['avg_ms'], 'blend_fwd', k['blend_fwd']['avg_ms'], 'psnr', d.get('psnr_vs_oracle_db'))"
done
