#!/bin/bash
# builds of the library with other register-ring depths in the small-batch / grouped int4 GEMM
# (kernel study): usage build_ring.sh "WD,XA,XAG" ... -> tools/exp/abl/libvla_ring_WD_XA_XAG.so
cd "$(dirname "$0")/../../vllm_legacy_amd/csrc" || exit 1
mkdir -p ../../tools/exp/abl
OTHERS=$(ls *.o | grep -v q4_gemm_fast.o)
for cfg in "$@"; do
  IFS=, read wd xa xag <<< "$cfg"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function \
    -DVLA_Q4_WD=$wd -DVLA_Q4_XA=$xa -DVLA_Q4_XAG=$xag -c q4_gemm_fast.hip -o /tmp/q4_ring_$$.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/q4_ring_$$.o \
    -o ../../tools/exp/abl/libvla_ring_${wd}_${xa}_${xag}.so || exit 1
done
