#!/bin/bash
# The profiling recipes of this repository, one script (on the GPU box, from anywhere; every path hangs off $ROOT):
#
#   bash tools/profile_round.sh <tag> [section ...]      sections (default: bench shapes c3 ranged)
#     bench     rocprofv3 --kernel-trace --stats of the default bench.py command and of the C3 headline alone
#               (the roofline's kernel must agree with the latter)
#     shapes    kernel statistics + FETCH_SIZE / WRITE_SIZE passes of a few module steps at C1, C2 and the C4 share
#     c3        FETCH / WRITE / clock passes of the scan and the filtered step at C3, kernel statistics of the latter
#     ranged    grouped searches of one C5 code shard (8 aligned roles, 5 unequal roles; scan and filtered): kernel
#               statistics + one PMC pass each, single-rank program (tools/run_kernels.py rangedN)
#     filter    the long SQ / LDS counter passes of the filtered search at C3 (tools/bench_filter.py)
#     tune=V,V  FETCH / WRITE passes of the distance kernel alone for tuning variants (T5VQ_TUNE)
#     timeline=<wl>:<search>   per launch duration and idle gap of the last module call (tools/trace_timeline.py)
#     cmd=<script.py>[:arg:arg]  kernel statistics of any python tool of this repository
# PMC passes are runs of their own (counters only, no tracing).  Summaries land in gpurun_out/profiles_<tag>/; copy
# what is to be judged into profiles/.
set -u
TAG=${1:-r3}; shift || true
SECTIONS=${*:-bench shapes c3 ranged}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

stats() {  # name, then the program and its arguments
  local name=$1; shift
  rm -rf "$OUT/trace_$name"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$name" -o t -- "$@" \
      > "$OUT/${name}_stdout.txt" 2> "$OUT/${name}_stderr.txt" || { echo "trace $name failed"; tail -3 "$OUT/${name}_stderr.txt"; return 1; }
  local f
  f=$(find "$OUT/trace_$name" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/${TAG}_${name}_kernel_stats.csv"
  python3 "$ROOT/tools/trace_by_grid.py" "$OUT/trace_$name" > "$OUT/${TAG}_${name}_kernels_by_grid.csv"
  echo "trace $name ok"
}
pmc_run() {  # name, counters..., --, program and arguments
  local name=$1; shift
  local ctr=()
  while [ "$1" != "--" ]; do ctr+=("$1"); shift; done
  shift
  rm -rf "$OUT/pmc_$name"
  timeout -s KILL 240 rocprofv3 --pmc "${ctr[@]}" --output-format csv -d "$OUT/pmc_$name" -- "$@" \
      > "$OUT/pmc_${name}.log" 2>&1 || { echo "pmc $name failed"; tail -3 "$OUT/pmc_${name}.log"; return 1; }
  echo "pmc $name ok"
}
pmc() {  # name, workload, counters...   (T5VQ_PROF_SEARCH=filter: the filtered step instead of the scan)
  local name=$1 wl=$2; shift 2
  pmc_run "$name" "$@" -- python3 "$ROOT/tools/run_kernels.py" "$wl" 3 step "${T5VQ_PROF_SEARCH:-scan}"
}
summary() {  # output file, then pmc pass names
  local f=$1; shift
  for p in "$@"; do python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_$p"; done > "$f" 2>&1
}

for S in $SECTIONS; do
case $S in
bench)
  stats bench python3 "$ROOT/bench.py" || exit 1
  grep "^{" "$OUT/bench_stdout.txt" | tail -1 > "$OUT/${TAG}_bench_under_rocprof.json"
  # the headline shape alone (no per-configuration table, no series point, no parity slice): every launch of the fused
  # distance + argmin kernel in this run is the C3 shape, so the per-symbol average is the roofline's kernel time
  stats c3_headline python3 "$ROOT/bench.py" --workload c3 --no-configs --no-cpu-baseline || exit 1
  grep "^{" "$OUT/c3_headline_stdout.txt" | tail -1 > "$OUT/${TAG}_c3_headline_under_rocprof.json"
  ;;
shapes)
  for wl in c1 c2 c4; do
    stats $wl python3 "$ROOT/tools/run_kernels.py" $wl 20 step scan || exit 1
    pmc ${wl}_fetch $wl FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES || exit 1
    pmc ${wl}_write $wl WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
    summary "$OUT/${TAG}_${wl}_pmc_summary.txt" ${wl}_fetch ${wl}_write
  done
  ;;
c3)
  pmc c3_fetch c3 FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES || exit 1
  pmc c3_write c3 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
  summary "$OUT/${TAG}_c3_pmc_summary.txt" c3_fetch c3_write
  # clocks and matrix-pipe occupancy in cycles (the fp16 filter runs power-limited: its time-based share of the nominal
  # peak understates what the pipe does per cycle), scan and filtered step; kernel statistics of the filtered step
  pmc c3_clock c3 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES || exit 1
  T5VQ_PROF_SEARCH=filter pmc c3_filter_clock c3 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE || exit 1
  T5VQ_PROF_SEARCH=filter pmc c3_filter_fetch c3 FETCH_SIZE || exit 1
  stats c3_filter python3 "$ROOT/tools/run_kernels.py" c3 20 step filter || exit 1
  summary "$OUT/${TAG}_c3_clocks_pmc_summary.txt" c3_clock c3_filter_clock c3_filter_fetch
  ;;
ranged)
  for rs in 8 5; do for search in scan filter; do
    n=ranged${rs}_$search
    stats $n python3 "$ROOT/tools/run_kernels.py" ranged$rs 20 search $search || exit 1
    pmc_run $n FETCH_SIZE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -- \
        python3 "$ROOT/tools/run_kernels.py" ranged$rs 3 search $search || exit 1
    { tail -1 "$OUT/${n}_stdout.txt"; python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_$n"; } > "$OUT/${TAG}_${n}_pmc_summary.txt" 2>&1
  done; done
  ;;
filter)
  F=(python3 "$ROOT/tools/bench_filter.py" c3)
  pmc_run filter_sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -- "${F[@]}" || exit 1
  pmc_run filter_sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM -- "${F[@]}" || exit 1
  pmc_run filter_tcc FETCH_SIZE TCC_HIT_sum -- "${F[@]}" || exit 1
  summary "$OUT/${TAG}_filter_pmc_summary.txt" filter_sq1 filter_sq2 filter_tcc
  ;;
tune=*)
  for V in $(echo "${S#tune=}" | tr , ' '); do
    export T5VQ_TUNE=$V
    D=(python3 "$ROOT/tools/run_kernels.py" c3 3 dist)
    pmc_run tune${V}_fetch FETCH_SIZE TCC_EA0_RDREQ_DRAM_sum -- "${D[@]}" || exit 1
    pmc_run tune${V}_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -- "${D[@]}" || exit 1
    summary "$OUT/${TAG}_tune${V}_pmc_summary.txt" tune${V}_fetch tune${V}_write
  done
  unset T5VQ_TUNE
  ;;
timeline=*)
  a=${S#timeline=}; wl=${a%%:*}; search=${a#*:}; what=step
  [ "$search" = dist ] && what=dist && search=scan
  stats tl_${wl}_$what python3 "$ROOT/tools/run_kernels.py" $wl 10 $what $search || exit 1
  python3 "$ROOT/tools/trace_timeline.py" "$OUT/trace_tl_${wl}_$what" 14 | tee "$OUT/${TAG}_timeline_${wl}_${what}.txt"
  ;;
cmd=*)
  IFS=: read -r -a A <<< "${S#cmd=}"
  name=$(basename "${A[0]}" .py)
  stats $name python3 "$ROOT/${A[0]}" "${A[@]:1}" || exit 1
  head -30 "$OUT/${TAG}_${name}_kernels_by_grid.csv"
  ;;
*) echo "unknown section $S"; exit 2 ;;
esac
done
rm -rf "$OUT"/trace_* "$OUT"/pmc_*/
ls "$OUT"
