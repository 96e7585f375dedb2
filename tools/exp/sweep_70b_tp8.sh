#!/bin/bash
# launch-shape sweep of the int4 GEMM at the Llama-3-70B TP-8 rank shapes, batch 256
# (VLA_Q4_TUNE="nw,splits,mt[,kw]", VLA_Q4_TUNE_KP=2: the K-part LDS-DMA kernel)
cd "$(dirname "$0")/../.."
run() {  # name tune [kp]
  out=$(VLA_Q4_TUNE="$2" VLA_Q4_TUNE_KP="$3" timeout -k 10 100 python3 tools/gemm_bench.py --model 70b --tp 8 --m 256 --iters 20 --sets 8 --only $1 2>&1 | tail -1)
  echo "$1 tune=$2 kp=$3 $(echo "$out" | sed 's/.*us_median": \([0-9.]*\).*TFLOPs": \([0-9.]*\).*/\1 us \2 TF/')"
}
for g in qkv o gate_up down; do
  run $g "" ""
  for t in 4,1,4 4,2,4 4,4,4 4,8,4 8,1,4 8,2,4 8,4,4 4,1,2 4,2,2 4,4,2 4,8,2 4,2,1 4,4,1 4,8,1; do run $g $t ""; done
  for t in 4,1,4 4,2,4 4,4,4; do run $g $t 2; done
done
