#!/bin/bash
# sweep row-tile height x K slices of the int4 GEMM at batch 256 over the per-GPU layer shapes of
# tensor-parallel degree $1 (default 8)
cd "$(dirname "$0")/../.."
tp=${1:-8}
for only in qkv o gate_up down; do
  echo -n "$only default: "
  timeout -k 10 60 python3 tools/gemm_bench.py --only $only --iters 30 --tp $tp 2>/dev/null | grep -o '"us_median": [0-9.]*' || exit 1
  for cfg in "4,1,4" "4,2,4" "4,4,4" "4,8,4" "4,1,2" "4,2,2" "4,4,2" "4,8,2" "4,1,1" "4,2,1" "4,4,1" "4,8,1"; do
    echo -n "$only nw,splits,mt=$cfg: "
    VLA_Q4_TUNE="$cfg" timeout -k 10 60 python3 tools/gemm_bench.py --only $only --iters 30 --tp $tp 2>/dev/null | grep -o '"us_median": [0-9.]*' || exit 1
  done
done
