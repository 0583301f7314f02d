# times tools/gi_tune.py with the default library and with every gpurun_in/lib_*.so (tuning builds)
set -e
cd $GRAFT_REPO_ROOT
python tools/gi_tune.py - 2>&1 | tail -1
for f in gpurun_in/lib_*.so; do echo $f; GIGS_LIB=$GRAFT_REPO_ROOT/$f python tools/gi_tune.py - 2>&1 | tail -1; done
python tools/gi_tune.py - 2>&1 | tail -1
