#!/bin/bash
# sweep launch shapes of the int4 GEMM over the small-N layer shapes (kernel study)
cd "$(dirname "$0")/../.."
timeout -k 10 120 python3 tools/gemm_bench.py --check || exit 1
for only in qkv o down; do
  for nw in 4 8; do
    for sp in 1 2 3 4 6 7 8; do
      echo -n "$only nw=$nw splits=$sp: "
      VLA_Q4_TUNE="$nw,$sp" timeout -k 10 60 python3 tools/gemm_bench.py --only $only --iters 20 2>/dev/null | grep -o '"us_median": [0-9.]*' || exit 1
    done
  done
done
