#!/bin/bash
# PMC passes for the hot kernels (one rocprofv3 run per pass; counters only, no tracing).
# usage (on the GPU box, from the repo root): bash tools/pmc_passes.sh <tag> <workload> <what>
set -u
TAG=${1:-r1}; WL=${2:-c3}; WHAT=${3:-step}
OUT=/root/repo/gpurun_out/pmc_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 /root/repo/tools/run_kernels.py $WL 3 $WHAT > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
  echo "pass $name ok"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE || exit 1
run fetch FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS || exit 1
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
run atom TCC_EA0_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_REQ_sum || exit 1
find $OUT -name "*counter_collection.csv" | head
