#!/bin/bash
# run the diagnostic builds (tools/exp/build_abl.sh) on one layer shape and print the stamps
cd "$(dirname "$0")/../.."
ONLY=${1:-gate_up}
shift
for mask in "$@"; do
  f=tools/exp/abl/libvla_abl$mask.so
  echo "== $f"
  VLA_LIB_PATH=$PWD/$f VLA_Q4_DMA=${DMA:-3} timeout -k 10 120 python3 tools/gemm_bench.py --m 256 --iters 20 --only $ONLY 2>&1 | grep "stamp" | tail -${TAIL:-14} || exit 1
done
