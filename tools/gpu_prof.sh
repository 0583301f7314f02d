# rocprofv3 kernel-trace summary of a bench command: tools/gpu_prof.sh TAG [bench args...]
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=$1; shift
mkdir -p gpurun_out/prof_${TAG}
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_${TAG} -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --repeats 1 "$@" > gpurun_out/prof_${TAG}.log 2>&1 || { tail -20 gpurun_out/prof_${TAG}.log; exit 1; }
f=$(find gpurun_out/prof_${TAG} -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:32]:
    print("%-70s calls %5s avg %9.1f us total %8.2f ms %5s%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
tail -1 gpurun_out/prof_${TAG}.log | cut -c1-200
