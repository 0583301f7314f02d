set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4c4
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -- python3 bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --repeats 1 > $O/prof_c4.log 2>&1 || { tail -20 $O/prof_c4.log; exit 1; }
cp $(find $O/prof_c4 -name "*kernel_stats.csv" | head -1) $O/c4_kernel_stats.csv
f=$(find $O/prof_c4 -name "*kernel_trace.csv" | head -1)
python tools/step_timeline.py $f --all-queues --min-us 0 --step 5 > $O/c4_step_timeline.txt
rm -rf $O/prof_c4
grep -n "blend_bwd\|preprocess_bwd\|specular_apply_multi_kernel<true>\|shade_bwd\|step length" $O/c4_step_timeline.txt | cut -c1-150
