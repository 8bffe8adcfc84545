#!/bin/bash
# FETCH_SIZE / WRITE_SIZE / DRAM read requests of the distance kernel for a list of tuning variants.
set -u
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
for V in "$@"; do
  export T5VQ_TUNE=$V
  OUT=/root/repo/gpurun_out/pmc_${TAG}_v$V
  mkdir -p $OUT
  timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_DRAM_sum --output-format csv -d $OUT/fetch -- python3 /root/repo/tools/run_kernels.py c3 3 dist > $OUT/fetch.log 2>&1 || { echo "fetch pass failed for $V"; tail -5 $OUT/fetch.log; exit 1; }
  timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/write -- python3 /root/repo/tools/run_kernels.py c3 3 dist > $OUT/write.log 2>&1 || { echo "write pass failed for $V"; exit 1; }
  echo "variant $V ok"
done
