#!/bin/bash
# Diagnostic builds of the library for tools/experiments/small_stamps.py: phase stamps + main-loop ablations.
# usage: tools/experiments/build_small_variants.sh "0 1 2 3 4 7 8"   (values of T5VQ_SMALL_ABL)
set -e
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
cd "$ROOT/t5vqvae_amd/csrc"
for abl in ${1:-0}; do
  /opt/rocm/bin/hipcc -DT5VQ_SMALL_STAMPS -DT5VQ_SMALL_ABL=$abl -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 \
    -I"$ROOT/include" -Wno-array-bounds -shared -Wl,-soname,libt5vq.so -o "$ROOT/tools/experiments/libt5vq_stamps_$abl.so" \
    t5vq_kernels.hip t5vq_filter.hip t5vq_small.hip &
done
wait
ls -la "$ROOT"/tools/experiments/libt5vq_stamps_*.so
