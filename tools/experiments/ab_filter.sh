#!/bin/bash
# Same-box A/B of two library builds on the filtered search: tools/experiments/libt5vq_A.so (baseline) against the
# in-tree library, alternating, plain (C3, C1, C4) and ranged shapes.   bash tools/experiments/ab_filter.sh
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
for round in 1 2; do
  for v in A B; do
    if [ $v = A ]; then export T5VQ_LIB=$ROOT/tools/experiments/libt5vq_A.so; else unset T5VQ_LIB; fi
    echo "== $v (round $round)"
    timeout -k 10 200 python3 "$ROOT/tools/bench_filter.py" c3 c4 c1 | grep -v amdgpu.ids || exit 1
    [ $round = 1 ] && { timeout -k 10 200 python3 "$ROOT/tools/experiments/ranged_filter_splits.py" 2>&1 | grep -v amdgpu.ids | sed 's/S=0.*S=4/S=4/; s/| S=8.*//' || exit 1; }
  done
done
