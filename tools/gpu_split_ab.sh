cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
timeout -k 10 300 python -m pytest tests/test_gpu_pbr.py -x -q > gpurun_out/r3d/pbr.log 2>&1; echo pbr rc=$?; tail -2 gpurun_out/r3d/pbr.log | cut -c1-200
for v in "1 0" "1 20" "1 60" "0 0"; do set -- $v
  GIGS_SHADE_BWD_SPLIT=$1 GIGS_SHADE_LIGHT_HEAD_START_US=$2 python bench.py --no-cpu-baseline --no-extras --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('split $1 head $2:', d['value'], d['repeats']['ms_per_step_median'])"
done
