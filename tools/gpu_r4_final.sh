# round-4 artefacts, part 1: GPU tests, rocprofv3 stats + timeline of the default command (C2) and of C4, bench lines -> gpurun_out/r4z/
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4z
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
tail -1 $O/pytest_gpu.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --repeats 1 > $O/prof_c2.log 2>&1 || { tail -20 $O/prof_c2.log; exit 1; }
f=$(find $O/prof_c2 -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f --all-queues --min-us 0 --step 12 > $O/c2_step_timeline.txt
cp $(find $O/prof_c2 -name "*kernel_stats.csv" | head -1) $O/c2_default_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -- python3 bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --repeats 1 > $O/prof_c4.log 2>&1 || { tail -20 $O/prof_c4.log; exit 1; }
cp $(find $O/prof_c4 -name "*kernel_stats.csv" | head -1) $O/c4_default_kernel_stats.csv
f=$(find $O/prof_c4 -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f --all-queues --min-us 0 --step 5 > $O/c4_step_timeline.txt
rm -rf $O/prof_c2 $O/prof_c4
echo "profiles done"
python bench.py > $O/bench_c2.json 2> $O/bench.err
echo "bench c2 done"
python bench.py --config c3 --steps 30 --warmup 5 > $O/bench_c3.json 2>> $O/bench.err
python bench.py --config c4 --steps 20 --warmup 5 --cpu-single-res 0 > $O/bench_c4.json 2>> $O/bench.err
echo "bench c4 done"
python bench.py --start 64 --no-cpu-baseline --no-extras > $O/bench_c2_start64_readme_setting.json 2>> $O/bench.err
for f in c2 c3 c4 c2_start64_readme_setting; do python -c "
import json
d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'], d.get('psnr_vs_oracle_db'), (d.get('iteration') or {}).get('iterations_per_s'), (d.get('iteration_cached_geometry') or {}).get('iterations_per_s'), (d.get('drop_in_step') or {}).get('renders_per_s'), (d.get('roofline') or {}).get('bound'), (d.get('roofline') or {}).get('frac'))"; done
