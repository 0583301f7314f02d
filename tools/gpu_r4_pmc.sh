# round-4 artefacts, part 2: PMC passes (separate rocprofv3 --pmc runs, --kernel-trace only) for C2 and C4 -> gpurun_out/r4z/
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r4z
mkdir -p $O
GIGS_PMC_TAG=r4z/pmc_c2 bash tools/gpu_pmc.sh > $O/pmc_c2.log 2>&1 || { tail -20 $O/pmc_c2.log; exit 1; }
python tools/pmc_summary.py $O/pmc_c2 > $O/pmc_summary_r04_c2.txt; cp $O/pmc_c2/summary.json $O/pmc_summary_r04_c2.json
echo "pmc c2 done"
GIGS_PMC_TAG=r4z/pmc_c4 GIGS_PMC_ARGS="--config c4" bash tools/gpu_pmc.sh > $O/pmc_c4.log 2>&1 || { tail -20 $O/pmc_c4.log; exit 1; }
python tools/pmc_summary.py $O/pmc_c4 > $O/pmc_summary_r04_c4.txt; cp $O/pmc_c4/summary.json $O/pmc_summary_r04_c4.json
rm -rf $O/pmc_c2 $O/pmc_c4
grep -n "ssao\|ssr_kernel" $O/pmc_summary_r04_c2.txt | head
