#!/bin/bash
# Kernel times of 20 module steps at one workload (rocprofv3 kernel trace, grouped by grid):
#   bash tools/experiments/trace_step.sh c3 scan
WL=${1:-c3}; SEARCH=${2:-scan}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/trace_${WL}_${SEARCH}
rm -rf "$OUT"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o t -- python3 "$ROOT/tools/run_kernels.py" "$WL" 20 step "$SEARCH" > "$OUT.log" 2>&1 || { tail -3 "$OUT.log"; exit 1; }
python3 "$ROOT/tools/trace_by_grid.py" "$OUT"
