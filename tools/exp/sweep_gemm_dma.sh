#!/bin/bash
# register-staged against LDS-DMA 128-row int4 GEMM (VLA_Q4_DMA: 0 off, 3 / 4 ring stages,
# 14 / 15 staggered waves), batch M (kernel study).  usage: sweep_gemm_dma.sh M mode...
cd "$(dirname "$0")/../.."
M=${1:-256}
shift
for dma in "$@"; do
  echo "== VLA_Q4_DMA=$dma m=$M"
  VLA_Q4_DMA=$dma timeout -k 10 120 python3 tools/gemm_bench.py --m $M --iters 40 --check 2>&1 | tail -1 || exit 1
done
