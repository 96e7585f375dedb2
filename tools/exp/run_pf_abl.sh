#!/bin/bash
# run the diagnostic builds (tools/exp/build_pf_abl.sh): bash tools/exp/run_pf_abl.sh <nseq> <seqlen> masks...
cd "$(dirname "$0")/../.."
NS=$1; SL=$2
shift 2
for mask in "$@"; do
  f=tools/exp/abl/libvla_pf$mask.so
  echo "== $f"
  VLA_LIB_PATH=$PWD/$f VLA_PF_FORM=2 PF_LAUNCHES=16 timeout -k 10 120 python3 tools/exp/prefill_stamp.py $NS $SL 2>&1 | grep "median" || exit 1
done
