#!/bin/bash
# sweep K slices across workgroups x K parts inside a workgroup for the small-batch int4 GEMM (kernel study)
cd "$(dirname "$0")/../.."
M=${1:-32}
MT=${2:-1}
timeout -k 10 120 python3 tools/gemm_bench.py --m $M --check > /dev/null || exit 1
for only in qkv o gate_up down; do
  for sp in 1 2 4 8; do
    for kw in 1 2 4; do
      echo -n "m=$M $only splits=$sp kw=$kw: "
      VLA_Q4_TUNE="4,$sp,$MT,$kw" timeout -k 10 60 python3 tools/gemm_bench.py --m $M --only $only --iters 40 --check 2>/dev/null | grep -o '"us_median": [0-9.]*' || exit 1
    done
  done
done
