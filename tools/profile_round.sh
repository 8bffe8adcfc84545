#!/bin/bash
# One round's profiling evidence (on the GPU box, from the repo root):  bash tools/profile_round.sh r2
#   - rocprofv3 --kernel-trace --stats of the default bench.py command (the roofline's kernel must agree with it)
#   - kernel statistics of a few module steps at C1, C2 and the C4 share (tools/run_kernels.py)
#   - PMC passes (FETCH_SIZE / WRITE_SIZE, one rocprofv3 run each: counters only, no tracing) for the same
# Summaries land in gpurun_out/profiles_<tag>/; copy what is to be judged into profiles/.
set -u
TAG=${1:-r2}
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
pmc() {  # name, workload, counters...   (T5VQ_PROF_SEARCH=filter: the filtered step instead of the scan)
  local name=$1 wl=$2; shift 2
  rm -rf "$OUT/pmc_$name"
  timeout -s KILL 240 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/pmc_$name" -- python3 "$ROOT/tools/run_kernels.py" "$wl" 3 step ${T5VQ_PROF_SEARCH:-scan} \
      > "$OUT/pmc_${name}.log" 2>&1 || { echo "pmc $name failed"; tail -3 "$OUT/pmc_${name}.log"; return 1; }
  echo "pmc $name ok"
}

stats bench python3 "$ROOT/bench.py" || exit 1
grep "^{" "$OUT/bench_stdout.txt" | tail -1 > "$OUT/${TAG}_bench_under_rocprof.json"
# the headline shape alone (no per-configuration table, no series point, no parity slice): every launch of the fused
# distance + argmin kernel in this run is the C3 shape, so the per-symbol average is the roofline's kernel time
stats c3_headline python3 "$ROOT/bench.py" --workload c3 --no-configs --no-cpu-baseline || exit 1
grep "^{" "$OUT/c3_headline_stdout.txt" | tail -1 > "$OUT/${TAG}_c3_headline_under_rocprof.json"
for wl in c1 c2 c4; do
  stats $wl python3 "$ROOT/tools/run_kernels.py" $wl 20 step scan || exit 1
  pmc ${wl}_fetch $wl FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES || exit 1
  pmc ${wl}_write $wl WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
done
pmc c3_fetch c3 FETCH_SIZE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES || exit 1
pmc c3_write c3 WRITE_SIZE TCC_HIT_sum TCC_MISS_sum || exit 1
# clocks and matrix-pipe occupancy in cycles (the fp16 filter runs power-limited: its time-based share of the nominal peak
# understates what the pipe does per cycle), scan and filtered step at C3; kernel statistics of the filtered step
pmc c3_clock c3 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES || exit 1
T5VQ_PROF_SEARCH=filter pmc c3_filter_clock c3 GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE || exit 1
T5VQ_PROF_SEARCH=filter pmc c3_filter_fetch c3 FETCH_SIZE || exit 1
stats c3_filter python3 "$ROOT/tools/run_kernels.py" c3 20 step filter || exit 1
{ echo "== c3 scan step: clocks =="; python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_c3_clock"; echo "== c3 filtered step: clocks, LDS, fetch =="; python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_c3_filter_clock"; python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_c3_filter_fetch"; } > "$OUT/${TAG}_c3_clocks_pmc_summary.txt" 2>&1
for wl in c1 c2 c3 c4; do
  { echo "== $wl: rocprofv3 --pmc, mean per launch (tools/pmc_summary.py) =="; python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_${wl}_fetch"; python3 "$ROOT/tools/pmc_summary.py" "$OUT/pmc_${wl}_write"; } > "$OUT/${TAG}_${wl}_pmc_summary.txt" 2>&1
done
rm -rf "$OUT"/trace_* "$OUT"/pmc_*/
ls "$OUT"
