# PMC passes over the SSAO / SSR kernels in three march configurations (exact, proj without / with certification)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/gi_pmc
mkdir -p $OUT
if [ $# -eq 0 ]; then set -- "exact 0" "proj 0" "proj 1"; fi
for cfg in "$@"; do
  set -- $cfg
  tag=$1_cert$2
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${tag}_a -- python3 tools/gi_pmc_driver.py $1 $2 > $OUT/${tag}_a.log 2>&1 || { tail -5 $OUT/${tag}_a.log; exit 1; }
  rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_WAVES --output-format csv -d $OUT/${tag}_b -- python3 tools/gi_pmc_driver.py $1 $2 > $OUT/${tag}_b.log 2>&1 || { tail -5 $OUT/${tag}_b.log; exit 1; }
  rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $OUT/${tag}_c -- python3 tools/gi_pmc_driver.py $1 $2 > $OUT/${tag}_c.log 2>&1 || echo "pass c failed for $tag (counter names)"
  echo "== $tag"
  python3 - $OUT $tag <<PY
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
for k in ("ssao_kernel", "ssr_kernel"):
    agg = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{tag}_*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if k in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(k, {c: round(sum(v) / len(v)) for c, v in sorted(agg.items())})
PY
done
