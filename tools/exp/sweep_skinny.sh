#!/bin/bash
# Kernel study: the small-batch int4 GEMM (q4_gemm_skinny.hip) against the 32-row kernel it
# replaces, and its launch shapes ("waves per workgroup, K slices") per layer shape.
#   bash tools/exp/sweep_skinny.sh [m]      -> gpurun_out/sweep_skinny_m<m>.log
m=${1:-16}
out=gpurun_out/sweep_skinny_m$m.log
: > $out
echo "== 32-row kernel" >> $out
VLA_Q4_SKINNY=0 timeout -k 10 120 python3 tools/gemm_bench.py --m $m --iters 20 >> $out 2>&1 || exit 1
echo "== skinny, default shapes" >> $out
timeout -k 10 120 python3 tools/gemm_bench.py --m $m --iters 20 >> $out 2>&1 || exit 1
echo "== skinny, two-chunk ring" >> $out
VLA_Q4_SKINNY_RING=2 timeout -k 10 120 python3 tools/gemm_bench.py --m $m --iters 20 >> $out 2>&1 || exit 1
for shape in qkv o gate_up down; do
  for tune in 4,1 8,1 16,1 4,2 8,2 16,2 4,4 8,4 16,4 8,7 8,8 16,8; do
    echo "== $shape $tune" >> $out
    VLA_Q4_SKINNY_TUNE=$tune timeout -k 10 120 python3 tools/gemm_bench.py --m $m --iters 20 --only $shape >> $out 2>&1 || exit 1
  done
done
