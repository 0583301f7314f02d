# GI march: default library vs every gpurun_in/lib_*.so (tools/gi_tune.py: SSAO + SSR on the C2 G-buffer, ms)
set -e
cd $GRAFT_REPO_ROOT
python tools/gi_tune.py - 2>&1 | tail -1
for f in gpurun_in/lib_*.so; do
  echo "== $f"
  GIGS_LIB=$GRAFT_REPO_ROOT/$f python tools/gi_tune.py - 2>&1 | tail -1
done
python tools/gi_tune.py - 2>&1 | tail -1
