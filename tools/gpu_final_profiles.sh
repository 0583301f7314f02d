# round-end artefacts: GPU tests, rocprofv3 stats (default / whole-step graph), PMC passes, bench lines (tag = $1)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-vX}
python -m pytest tests -m gpu -x -q > gpurun_out/pytest_${TAG}.log 2>&1 || { tail -20 gpurun_out/pytest_${TAG}.log; exit 1; }
tail -1 gpurun_out/pytest_${TAG}.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_${TAG}.log 2>&1 || { tail -20 gpurun_out/prof_${TAG}.log; exit 1; }
f=$(find gpurun_out/prof_${TAG} -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f --all-queues --min-us 0 --step 8 > gpurun_out/${TAG}_c2_step_timeline.txt
cp $(find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1) gpurun_out/${TAG}_c2_kernel_stats.csv
rm -rf gpurun_out/pmc
bash tools/gpu_pmc.sh > gpurun_out/pmc_run.log 2>&1 || { tail -20 gpurun_out/pmc_run.log; exit 1; }
python tools/pmc_summary.py gpurun_out/pmc > gpurun_out/${TAG}_pmc_summary.txt
python bench.py --steps 30 --warmup 5 > gpurun_out/bench_${TAG}_c2.json 2> gpurun_out/bench_${TAG}.err
python bench.py --config c3 --steps 30 --warmup 5 > gpurun_out/bench_${TAG}_c3.json 2>> gpurun_out/bench_${TAG}.err
python bench.py --config c4 --steps 10 --warmup 3 > gpurun_out/bench_${TAG}_c4.json 2>> gpurun_out/bench_${TAG}.err
GIGS_RASTER_GRAPH=1 python bench.py --no-cpu-baseline > gpurun_out/bench_${TAG}_c2_rastergraph.json 2>> gpurun_out/bench_${TAG}.err
GIGS_STEP_GRAPH=0 python bench.py --no-cpu-baseline > gpurun_out/bench_${TAG}_c2_eager_raster.json 2>> gpurun_out/bench_${TAG}.err
python bench.py --start 64 --no-cpu-baseline > gpurun_out/bench_${TAG}_c2_start64.json 2>> gpurun_out/bench_${TAG}.err
for f in c2 c3 c4 c2_rastergraph c2_eager_raster c2_start64; do python -c "
import json
d=json.loads(open('gpurun_out/bench_${TAG}_$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'], d.get('psnr_vs_oracle_db'))"; done
