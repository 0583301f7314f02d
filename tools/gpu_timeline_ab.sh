# timelines of one timed step per step mode: $2.. = "NAME:ENV=VAL,ENV=VAL" (default: whole-step graph vs eager rasterizer)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-tl}
shift || true
MODES=${@:-"whole:GIGS_STEP_GRAPH=1 eager:GIGS_STEP_GRAPH=0"}
for m in $MODES; do
  name=${m%%:*}
  envs=$(echo ${m#*:} | tr ',' ' ')
  rm -rf gpurun_out/prof_${TAG}_$name
  for e in $envs; do export $e; done
  rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_${TAG}_$name -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/prof_${TAG}_$name.log 2>&1 || { tail -20 gpurun_out/prof_${TAG}_$name.log; exit 1; }
  f=$(find gpurun_out/prof_${TAG}_$name -name "*kernel_trace.csv" | head -1)
  python tools/step_timeline.py $f --all-queues --min-us 0 --step 8 > gpurun_out/${TAG}_timeline_$name.txt
  rm -rf gpurun_out/prof_${TAG}_$name
done
