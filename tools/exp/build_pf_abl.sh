#!/bin/bash
# diagnostic builds of the library: the 512-thread prompt attention kernel (fp16, head 128, no
# prefix only: -DVLA_PF_QUICK) with parts left out (VLA_PF_ABL bit mask: 1 softmax arithmetic,
# 2 second product, 4 staging loads and LDS stores, 8 barriers, 16 first product)
# -> tools/exp/abl/libvla_pf<mask>.so, used through VLA_LIB_PATH.  Timing studies only.
cd "$(dirname "$0")/../../vllm_legacy_amd/csrc" || exit 1
mkdir -p ../../tools/exp/abl
OTHERS=$(ls *.o | grep -v prefill_attention.o)
for mask in "$@"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function \
    -DVLA_PF_QUICK -DVLA_PF_ABL=$mask -c prefill_attention.hip -o /tmp/pf_abl_$mask.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS /tmp/pf_abl_$mask.o \
    -o ../../tools/exp/abl/libvla_pf$mask.so || exit 1
done
