#!/bin/bash
# K-sliced layers at batch 256: 128 x 32 wave tiles (VLA_Q4_DMA=3) against 128 x 64 wave tiles with two
# K parts per workgroup (23) at a cap on the K slices across workgroups (kernel study)
cd "$(dirname "$0")/../.."
for only in qkv o down; do
  for dma in 3 23; do
    for cap in 2 4 8; do
      echo -n "$only dma=$dma maxsplit=$cap: "
      VLA_Q4_DMA=$dma VLA_Q4_MAXSPLIT=$cap timeout -k 10 60 python3 tools/gemm_bench.py --m 256 --only $only --iters 40 2>/dev/null | grep -o '"us_median": [0-9.]*' || exit 1
    done
  done
done
