# round-3 artefacts: GPU tests, rocprofv3 stats + timeline of the default command, PMC passes, bench lines -> gpurun_out/r3f/
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r3f
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -40 $O/pytest_gpu.log; exit 1; }
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
GIGS_PMC_TAG=r3f/pmc_c2 bash tools/gpu_pmc.sh > $O/pmc_c2.log 2>&1 || { tail -20 $O/pmc_c2.log; exit 1; }
python tools/pmc_summary.py $O/pmc_c2 > $O/pmc_summary_r03_c2.txt; cp $O/pmc_c2/summary.json $O/pmc_summary_r03_c2.json
GIGS_PMC_TAG=r3f/pmc_c4 GIGS_PMC_ARGS="--config c4" bash tools/gpu_pmc.sh > $O/pmc_c4.log 2>&1 || { tail -20 $O/pmc_c4.log; exit 1; }
python tools/pmc_summary.py $O/pmc_c4 > $O/pmc_summary_r03_c4.txt; cp $O/pmc_c4/summary.json $O/pmc_summary_r03_c4.json
rm -rf $O/pmc_c2 $O/pmc_c4
python bench.py > $O/bench_c2.json 2> $O/bench.err
python bench.py --config c3 --steps 30 --warmup 5 > $O/bench_c3.json 2>> $O/bench.err
python bench.py --config c4 --steps 20 --warmup 5 --cpu-single-res 0 > $O/bench_c4.json 2>> $O/bench.err
python bench.py --start 64 --no-cpu-baseline --no-extras > $O/bench_c2_start64_readme_setting.json 2>> $O/bench.err
GIGS_STEP_GRAPH=0 python bench.py --no-cpu-baseline --no-extras > $O/bench_c2_eager_rasterizer.json 2>> $O/bench.err
GIGS_BENCH_FORCE_DIST=1 python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_c5_1rank_rccl_rehearsal.json 2>> $O/bench.err
GIGS_BENCH_FORCE_DIST=1 python bench.py --no-cpu-baseline --no-extras > $O/bench_c2_1rank_rccl_rehearsal.json 2>> $O/bench.err
GIGS_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_c2_2rank_gloo_selflaunch.json 2>> $O/bench.err
for f in c2 c3 c4 c2_start64_readme_setting c2_eager_rasterizer c5_1rank_rccl_rehearsal c2_1rank_rccl_rehearsal c2_2rank_gloo_selflaunch; do python -c "
import json
d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'], d.get('psnr_vs_oracle_db'), d.get('n_gpus'), d.get('ranks_seen'), (d.get('iteration') or {}).get('iterations_per_s'))"; done
