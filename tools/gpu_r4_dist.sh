# round 4: one-GPU rehearsals of the multi-rank bench incl. the complete iterations under data parallelism
# (1) one rank over RCCL at C2 and C5 (both reduce modes of the metric's step; DP iterations with `comm`), (2) two ranks over gloo
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4d
mkdir -p $O
python -m pytest tests/test_gpu_training.py -q -k "gradient_slab" > $O/slab_tests.log 2>&1 || { tail -40 $O/slab_tests.log; exit 1; }
tail -1 $O/slab_tests.log
GIGS_BENCH_FORCE_DIST=1 timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --iteration --no-extras > $O/bench_c2_1rank_rccl_rehearsal.json 2> $O/bench_c2_1rank.err || { tail -40 $O/bench_c2_1rank.err; exit 1; }
GIGS_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --iteration --no-extras > $O/bench_c2_2rank_gloo_selflaunch.json 2> $O/bench_c2_2rank.err || { tail -40 $O/bench_c2_2rank.err; exit 1; }
# the driver's own launch line (torchrun), two ranks sharing the card over gloo
GIGS_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29571 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $O/bench_c2_2rank_gloo_torchrun.json 2> $O/bench_c2_2rank_torchrun.err || { tail -40 $O/bench_c2_2rank_torchrun.err; exit 1; }
tail -c 300 $O/bench_c2_2rank_gloo_torchrun.json; echo
GIGS_BENCH_FORCE_DIST=1 timeout -k 10 500 python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline --iteration --no-extras > $O/bench_c5_1rank_rccl_rehearsal.json 2> $O/bench_c5_1rank.err || { tail -40 $O/bench_c5_1rank.err; exit 1; }
GIGS_BENCH_FORCE_DIST=1 GIGS_BENCH_REDUCE=all timeout -k 10 400 python bench.py --config c5 --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_c5_1rank_rccl_rehearsal_reduce_all.json 2>> $O/bench_c5_1rank.err || { tail -40 $O/bench_c5_1rank.err; exit 1; }
for f in c2_1rank_rccl_rehearsal c2_2rank_gloo_selflaunch c5_1rank_rccl_rehearsal c5_1rank_rccl_rehearsal_reduce_all; do python -c "
import json
d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1])
it=d.get('iteration') or {}; s1=d.get('iteration_stage1') or {}
print('$f', d['value'], d['ms_per_step'], d.get('n_gpus'), d.get('ranks_seen'), d.get('comm'))
print('   iteration', it.get('iterations_per_s'), it.get('error'), (it.get('data_parallel') or {}).get('comm'))
print('   stage1   ', s1.get('iterations_per_s'), s1.get('error'), (s1.get('data_parallel') or {}).get('comm'))"; done
