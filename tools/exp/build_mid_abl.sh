#!/bin/bash
# diagnostic builds of the library: the 17..80-row int4 GEMM with one thing removed each
# (VLA_MID_ABL bit mask, see q4_gemm_mid.hip) -> tools/exp/abl/libvla_mid<mask>.so, used through
# VLA_LIB_PATH.  Timing studies only: ablated builds compute wrong results.
cd "$(dirname "$0")/../../vllm_legacy_amd/csrc" || exit 1
mkdir -p ../../tools/exp/abl
OTHERS=$(ls *.o | grep -v q4_gemm_mid.o)
for mask in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function \
    -DVLA_MID_ABL=$mask -c q4_gemm_mid.hip -o /tmp/q4_mid_$mask.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/q4_mid_$mask.o \
    -o ../../tools/exp/abl/libvla_mid$mask.so || exit 1
done
