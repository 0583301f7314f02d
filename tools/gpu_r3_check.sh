# round-3 check: GPU tests, the default bench line, the self-launched 2-rank gloo rehearsal, the one-rank RCCL c5 rehearsal
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
python -m pytest tests -m gpu -x -q > gpurun_out/r3b/gpu_tests.log 2>&1 || { tail -80 gpurun_out/r3b/gpu_tests.log; exit 1; }
tail -3 gpurun_out/r3b/gpu_tests.log
timeout -k 10 500 python bench.py > gpurun_out/r3b/bench_c2.json 2> gpurun_out/r3b/bench_c2.err || { tail -40 gpurun_out/r3b/bench_c2.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3b/bench_c2.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['repeats'])
for k in ('drop_in_step','full_grad_step','iteration'):
    print(k, d.get(k))
print(d['cpu_baseline'])
print(json.dumps(d['parity_c2']['gi_per_pixel'])[:3000])
print({k:d['config'].get(k) for k in ('K_pairs_evaluated','K_pairs_contributing','covered_px_frac')})
PY
GIGS_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 > gpurun_out/r3b/bench_2rank_gloo.json 2> gpurun_out/r3b/bench_2rank.err || { tail -40 gpurun_out/r3b/bench_2rank.err; exit 1; }
cut -c1-300 gpurun_out/r3b/bench_2rank_gloo.json
python -c "
import json
d=json.loads(open('gpurun_out/r3b/bench_2rank_gloo.json').read().strip())
print(d['n_gpus'], d['ranks_seen'], d['comm'])"
GIGS_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --config c5 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/r3b/bench_c5_1rank_rccl.json 2> gpurun_out/r3b/bench_c5.err || { tail -40 gpurun_out/r3b/bench_c5.err; exit 1; }
python -c "
import json
d=json.loads(open('gpurun_out/r3b/bench_c5_1rank_rccl.json').read().strip())
print(d['value'], d['n_gpus'], d['ranks_seen'], d['comm'])"
