set -e
cd $GRAFT_REPO_ROOT
python tools/gi_tune.py - 2>&1 | tail -1
for v in g8_r1 g2_r4 g4_r4 g8_r2 g2_r2; do
  echo $v; GIGS_LIB=$GRAFT_REPO_ROOT/gpurun_in/lib_$v.so python tools/gi_tune.py - 2>&1 | tail -1
done
python tools/gi_tune.py - 2>&1 | tail -1
