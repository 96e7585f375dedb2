#!/bin/bash
# Kernel study: chunk length (K steps per zero/scale fetch and ring slot) of the small-batch int4
# GEMM against launch shapes.   bash tools/exp/sweep_skinny_cs.sh [m]  -> gpurun_out/sweep_skinny_cs_m<m>.log
m=${1:-16}
out=gpurun_out/sweep_skinny_cs_m$m.log
: > $out
for cs in 4 2 1; do
  for shape in qkv o gate_up down; do
    for tune in 4,1 8,1 16,1 8,2 16,2 4,4 8,4 16,4; do
      echo "== cs $cs $shape $tune" >> $out
      VLA_Q4_SKINNY_CS=$cs VLA_Q4_SKINNY_TUNE=$tune timeout -k 10 120 python3 tools/gemm_bench.py --m $m --iters 20 --only $shape 2>&1 | grep layer_us >> $out || exit 1
    done
  done
done
