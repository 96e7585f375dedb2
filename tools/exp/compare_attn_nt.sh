#!/bin/bash
# nontemporal against default-policy K/V loads in the MFMA paged-attention kernel (kernel study):
# tools/exp/abl/libvla_kvdef.so is the library built with -DVLA_KV_DEFAULT_POLICY
cd "$(dirname "$0")/../.."
for args in "" "--kv auto" "--block 32" "--ragged" "--ragged --order" "--version v2" "--batch 16 --version v2" "--batch 64 --seq 16384" "--pool 2048" "--head-size 64" "--batch 32 --seq 1024"; do
  for lib in ${LIBS:-vllm_legacy_amd/libvla_gfx950.so tools/exp/abl/libvla_kvdef.so}; do
    echo -n "[$args] $(basename $lib): "
    VLA_LIB_PATH=$PWD/$lib timeout -k 10 100 python3 tools/attn_bench.py --iters 20 $args 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_median'], d['GBps_median'])" || exit 1
  done
done
