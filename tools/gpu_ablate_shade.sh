# shade backward: where the specular light-gradient scatter spends its time (GIGS_ABLATE bits: 1 diffuse, 2 all
# specular, 4 specular levels accumulated in LDS, 8 specular levels added with global atomics)
set -e
cd $GRAFT_REPO_ROOT
for ab in 0 2 4 8; do
  GIGS_ABLATE=$ab python bench.py --steps 10 --warmup 3 --no-cpu-baseline --graphs off > gpurun_out/bench_abl.json 2> gpurun_out/bench_abl.err || { tail -20 gpurun_out/bench_abl.err; exit 1; }
  python -c "
import json
d=json.loads(open('gpurun_out/bench_abl.json').read().strip().splitlines()[-1])
print('ablate=$ab shade_bwd', d['kernels']['shade_bwd']['ms_per_step'])"
done
