#!/bin/bash
# sweep launch shapes of the int4 GEMM over the per-GPU layer shapes at TP 2/4/8 (kernel study)
cd "$(dirname "$0")/../.."
for tp in 2 4 8; do
  echo "== tp=$tp default"; timeout -k 10 60 python3 tools/gemm_bench.py --tp $tp --iters 20 --sets 4 2>/dev/null | tail -1
  for only in qkv o gate_up down; do
    for nw in 4 8; do
      for sp in 1 2 4 8 16; do
        echo -n "tp=$tp $only nw=$nw splits=$sp: "
        VLA_Q4_TUNE="$nw,$sp" timeout -k 10 60 python3 tools/gemm_bench.py --tp $tp --only $only --iters 20 --sets 4 2>/dev/null | grep -o '"us_median": [0-9.]*' || exit 1
      done
    done
  done
done
