set -e
cd $GRAFT_REPO_ROOT
GIGS_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_2rank_gloo.json 2> gpurun_out/bench_2rank.err || { tail -40 gpurun_out/bench_2rank.err; exit 1; }
tail -1 gpurun_out/bench_2rank_gloo.json | cut -c1-300
