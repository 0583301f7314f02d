# bin_scatter in bands of tile rows (gigs_options.bin_bands / GIGS_BIN_BANDS) at C4: parity of the dense path, then the step and
# the scatter + sort stage per band count
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4i
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -q -m gpu -k "dense or binning_paths or c4_forward or identical_depths or async_binning" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for b in 1 2 4 8 16; do
  GIGS_BIN_BANDS=$b timeout -k 10 200 python bench.py --config c4 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --repeats 3 > $O/bench_c4_bands$b.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
  python -c "
import json
d=json.loads(open('$O/bench_c4_bands$b.json').read().strip().splitlines()[-1])
print('bands $b', d['value'], d['ms_per_step'], d['repeats']['ms_per_step_median'], 'sort stage (scatter + sorts)', d['kernels']['sort']['avg_ms'], 'count+prefix', d['kernels']['duplicate']['avg_ms'])"
done
