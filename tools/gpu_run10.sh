set -e
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for e in 1 0; do
  GIGS_MIPS_EARLY=$e python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_e$e.json 2> gpurun_out/bench_e.err || { tail -30 gpurun_out/bench_e.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/bench_e$e.json').read().strip().splitlines()[-1])
print('early=$e', d['value'], d['ms_per_step'])"
done
done
