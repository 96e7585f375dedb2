#!/bin/bash
# run the diagnostic builds (tools/exp/build_mid_abl.sh): bash tools/exp/run_mid_abl.sh <tokens> masks...
# prints the MoE block time from a graph for each build
cd "$(dirname "$0")/../.."
TOK=$1
shift 1
for mask in "$@"; do
  f=tools/exp/abl/libvla_mid$mask.so
  echo "== $f tokens=$TOK"
  VLA_LIB_PATH=$PWD/$f timeout -k 10 120 python3 tools/moe_bench.py --tokens $TOK --grouped-only 2>&1 | grep grouped_graph_ms || exit 1
done
