#!/bin/bash
# diagnostic builds of the library: the int4 GEMM with in-kernel clock stamps and one ablation each
# (VLA_Q4_ABL bit mask, see q4_gemm_fast.hip) -> tools/exp/abl/libvla_abl<mask>.so, used through
# VLA_LIB_PATH.  Timing studies only: ablated builds compute wrong results.
cd "$(dirname "$0")/../../vllm_legacy_amd/csrc" || exit 1
mkdir -p ../../tools/exp/abl
OTHERS=$(ls *.o | grep -v q4_gemm_fast.o)
for mask in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -DVLA_Q4_STAMP \
    -DVLA_Q4_ABL=$mask -c q4_gemm_fast.hip -o /tmp/q4_abl_$mask.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/q4_abl_$mask.o \
    -o ../../tools/exp/abl/libvla_abl$mask.so || exit 1
done
