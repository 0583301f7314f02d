# rocprofv3 kernel-trace summary of the complete-iteration leg: tools/gpu_prof_iter.sh TAG [bench args...]
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
mkdir -p gpurun_out/prof_${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --iteration --repeats 1 "$@" > gpurun_out/prof_${TAG}.log 2>&1 || { tail -20 gpurun_out/prof_${TAG}.log; exit 1; }
f=$(find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:60]:
    print("%-90s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
tail -1 gpurun_out/prof_${TAG}.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('iteration'), d.get('iteration_stage1'))"
