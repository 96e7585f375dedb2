#!/bin/bash
# 32-row against 64-row tiles of the small-batch int4 kernel at 33..64 rows (Llama-3-8B layer
# shapes; VLA_Q4_TUNE="nw,splits,mt,kw")
cd "$(dirname "$0")/../.."
for m in 48 64; do
  for g in qkv o gate_up down; do
    for t in "" 4,1,1,4 4,1,1,2 4,2,1,2 4,4,1,2 4,8,1,2 4,2,1,4 4,4,1,1 4,1,2,4 4,2,2,2 4,4,2,2 4,8,2,2 4,2,2,4; do
      out=$(VLA_Q4_TUNE="$t" timeout -k 10 100 python3 tools/gemm_bench.py --m $m --iters 20 --sets 8 --only $g 2>&1 | tail -1)
      echo "m=$m $g tune=$t $(echo "$out" | sed 's/.*us_median": \([0-9.]*\).*/\1 us/')"
    done
  done
done
