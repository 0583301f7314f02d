# kernel statistics of the complete-iteration leg (stage 2, stage 2 with the geometry cache, stage 1) at C2 and C4 -> gpurun_out/r4i/
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4i
mkdir -p $O
for c in c2 c4; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$c -- python3 bench.py --config $c --no-cpu-baseline --no-extras --iteration --repeats 1 --steps 10 --warmup 3 > $O/prof_$c.log 2>&1 || { tail -20 $O/prof_$c.log; exit 1; }
  cp $(find $O/prof_$c -name "*kernel_stats.csv" | head -1) $O/${c}_iteration_kernel_stats.csv
  rm -rf $O/prof_$c
  python3 - $O/${c}_iteration_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:14]:
    print("%-70s calls %5s avg %9.1f us total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
