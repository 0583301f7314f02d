set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
timeout -k 10 500 python bench.py --config c4 --steps 20 --warmup 5 --cpu-single-res 0 > gpurun_out/r3c/bench_c4.json 2> gpurun_out/r3c/bench_c4.err || { tail -30 gpurun_out/r3c/bench_c4.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3c/bench_c4.json').read().strip().splitlines()[-1])
print("C4", d['value'], d['ms_per_step'], d['repeats']['ms_per_step_median'], d['config']['rasterizer'], d['config']['R'])
print({k:v['ms_per_step'] for k,v in d['kernels'].items()})
p=d['parity_c4']
print({k:p[k] for k in ('num_rendered','radii_equal','keys_equal','point_list_equal','ranges_equal','n_contrib_flips','worst_plane_mean_l1','psnr_render_rgb')})
print(d.get('iteration'), d.get('drop_in_step'))
PY
GIGS_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r3c/bench_c5_1rank_rccl.json 2> gpurun_out/r3c/bench_c5.err || { tail -30 gpurun_out/r3c/bench_c5.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/r3c/bench_c5_1rank_rccl.json').read().strip())
print('C5 1 rank rccl', d['value'], d['n_gpus'], d['ranks_seen'], d['comm'], d['config']['rasterizer'])"
GIGS_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r3c/bench_2rank_gloo.json 2> gpurun_out/r3c/bench_2rank.err || { tail -30 gpurun_out/r3c/bench_2rank.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/r3c/bench_2rank_gloo.json').read().strip())
print('gloo x2', d['value'], d['n_gpus'], d['ranks_seen'], d['comm'])"
