# one-GPU rehearsals of the distributed bench path: (1) a single-rank RCCL process group with hipGraphs,
# (2) two ranks sharing the card over gloo
set -e
cd $GRAFT_REPO_ROOT
GIGS_BENCH_FORCE_DIST=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_1rank_rccl.json 2> gpurun_out/bench_1rank.err || { tail -40 gpurun_out/bench_1rank.err; exit 1; }
tail -1 gpurun_out/bench_1rank_rccl.json | cut -c1-220
GIGS_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank.err || { tail -40 gpurun_out/bench_2rank.err; exit 1; }
tail -1 gpurun_out/bench_2rank_gloo.json | cut -c1-220
GIGS_BENCH_FORCE_DIST=1 GIGS_BENCH_REDUCE=trainable timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_1rank_rccl_trainable.json 2>> gpurun_out/bench_1rank.err || { tail -40 gpurun_out/bench_1rank.err; exit 1; }
tail -1 gpurun_out/bench_1rank_rccl_trainable.json | cut -c1-220
