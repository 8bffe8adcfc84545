#!/bin/bash
# PMC passes for the filtered search (one rocprofv3 run per pass; counters only).  usage: bash tools/pmc_filter.sh <tag>
set -u
TAG=${1:-filter}
OUT=/root/repo/gpurun_out/pmc_${TAG}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  local name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 /root/repo/tools/bench_filter.py c3 > $OUT/$name.log 2>&1 || { echo "pass $name failed"; tail -5 $OUT/$name.log; exit 1; }
  echo "pass $name ok"
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE || exit 1
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM || exit 1
run tcc FETCH_SIZE TCC_HIT_sum || exit 1
