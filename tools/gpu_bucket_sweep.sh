# dense-scene binning: bucket-size sweep on C4 (rocprofv3 kernel stats of the binning kernels)
cd $GRAFT_REPO_ROOT
for tgt in 1536 3072 6144; do
  echo "== target $tgt"
  GIGS_BUCKET_TARGET=$tgt bash tools/gpu_prof.sh c4_t$tgt --config c4 --steps 10 --warmup 3 2>&1 | grep "long_\|bin_\|renders" | cut -c1-130
  tail -1 gpurun_out/prof_c4_t$tgt.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('renders/s', d['value'])" 2>/dev/null
done
