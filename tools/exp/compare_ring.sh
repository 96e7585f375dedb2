#!/bin/bash
# ring-depth builds (tools/exp/build_ring.sh) on the small-batch int4 GEMM and the MoE block
cd "$(dirname "$0")/../.."
for lib in "$@"; do
  for m in 16 64; do
    echo -n "$(basename $lib) m=$m: "
    VLA_LIB_PATH=$PWD/$lib timeout -k 10 120 python3 tools/gemm_bench.py --m $m --iters 40 --check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:v['us_median'] for k,v in d.items() if isinstance(v,dict) and 'us_median' in v}, d.get('layer_us'))" || exit 1
  done
  echo -n "$(basename $lib) moe: "
  VLA_LIB_PATH=$PWD/$lib timeout -k 10 120 python3 tools/moe_bench.py --tokens 16,64,256,1024 --grouped-only 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({r['tokens']: r['grouped_graph_ms'] for r in d})" || exit 1
done
