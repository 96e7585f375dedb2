#!/bin/bash
# Kernel study: up to how many rows per expert the grouped small-batch kernel (16-row tiles on
# grid.z) beats the LDS-staged tiles in the Mixtral block.  -> gpurun_out/sweep_moe_rows.log
out=gpurun_out/sweep_moe_rows.log
: > $out
for rows in 16 32 48 64; do
  echo "== VLA_Q4_SKINNY_ROWS=$rows" >> $out
  VLA_Q4_SKINNY_ROWS=$rows timeout -k 10 150 python3 tools/moe_bench.py --tokens 16,17,24,32,40,48,64 --grouped-only 2>/dev/null >> $out || exit 1
done
