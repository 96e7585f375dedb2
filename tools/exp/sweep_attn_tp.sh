#!/bin/bash
# decode attention at the per-GPU head counts of tensor-parallel ranks (batch 256 x 4096 tokens,
# fp8 cache): one workgroup per sequence and KV head (V1) against 512-token partitions (V2)
cd "$(dirname "$0")/../.."
for heads in "16 4" "8 2" "4 1" "8 1"; do
  set -- $heads
  for v in v1 v2; do
    echo -n "hq=$1 hkv=$2 $v: "
    timeout -k 10 120 python3 tools/attn_bench.py --hq $1 --hkv $2 --version $v --graph --layers 8 2>/dev/null | grep -o '"ms_median": [0-9.]*, "ms_min": [0-9.]*' || exit 1
  done
done
