#!/bin/bash
# usage (GPU box): ABLS="0 3" FLAGS="-DFWN_SETPRIO=1" tools/run_bench_gemm.sh [B]
cd "$(dirname "$0")/.."
for abl in ${ABLS:-0 3}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DFWN_ABL=$abl $FLAGS tools/bench_gemm.hip -o /tmp/bench_gemm_$abl 2>&1 | grep -E "error" -A3
  echo "=== ABL=$abl FLAGS=$FLAGS"
  /tmp/bench_gemm_$abl ${1:-8}
done
