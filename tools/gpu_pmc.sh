# Collects PMC counters for the hot kernels: separate rocprofv3 --pmc passes with --kernel-trace only.
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${GIGS_PMC_TAG:-pmc}
mkdir -p $OUT
run() { # name counters...
  name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --repeats 1 $GIGS_PMC_ARGS > $OUT/$name.log 2>&1 || { tail -20 $OUT/$name.log; exit 1; }
  echo "pass $name done"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY
run sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_WAVES
run fetch FETCH_SIZE GRBM_GUI_ACTIVE
run write WRITE_SIZE TCC_EA0_ATOMIC_sum
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
find $OUT -name "*counter_collection.csv" | head
