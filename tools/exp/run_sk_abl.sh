#!/bin/bash
# run the diagnostic builds (tools/exp/build_sk_abl.sh): bash tools/exp/run_sk_abl.sh <m> <shape> <tune> masks...
cd "$(dirname "$0")/../.."
M=$1; ONLY=$2; TUNE=$3
shift 3
for mask in "$@"; do
  f=tools/exp/abl/libvla_sk$mask.so
  echo "== $f $ONLY $TUNE m=$M"
  VLA_LIB_PATH=$PWD/$f VLA_Q4_SKINNY_TUNE=$TUNE timeout -k 10 120 python3 tools/gemm_bench.py --m $M --iters 20 --only $ONLY 2>&1 | grep layer_us || exit 1
done
