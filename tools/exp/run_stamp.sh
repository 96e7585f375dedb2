#!/bin/bash
# print the in-kernel time stamps of the diagnostic build (tools/exp/build_abl.sh 0) for one layer
# shape at batch M; usage: run_stamp.sh M shape [lines]
cd "$(dirname "$0")/../.."
VLA_LIB_PATH=$PWD/tools/exp/abl/libvla_abl0.so timeout -k 10 120 python3 tools/gemm_bench.py --m ${1:-16} --iters 8 --only ${2:-o} 2>&1 | grep "stamp" | tail -${3:-12}
