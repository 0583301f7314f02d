cd $GRAFT_REPO_ROOT
for il in 0 1; do
  GIGS_GI_INTERLEAVE=$il python bench.py --no-cpu-baseline --no-extras --repeats 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('interleave $il:', d['value'], d['repeats']['ms_per_step_median'], 'ssao', k['ssao']['avg_ms'], 'ssr', k['ssr']['avg_ms'])"
done
GIGS_GI_INTERLEAVE=1 timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -k "gi_" 2>&1 | tail -2
