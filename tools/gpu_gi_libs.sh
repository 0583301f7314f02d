# GI march: default library vs every gpurun_in/lib_*.so -- speed (tools/gi_tune.py); accuracy vs the exact march for the default
set -e
cd $GRAFT_REPO_ROOT
python tools/gi_tune.py - 2>&1 | tail -1
for f in gpurun_in/lib_*.so; do
  echo "== $f"
  GIGS_LIB=$GRAFT_REPO_ROOT/$f python tools/gi_tune.py - 2>&1 | tail -1
done
python tools/gi_tune.py - 2>&1 | tail -1
python tools/gi_variants.py --modes proj --no-sweep --reps 5 2>&1 | grep -v "^$" | tail -24
