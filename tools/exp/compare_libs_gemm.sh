#!/bin/bash
# two builds of the library on the int4 GEMM at batch M (kernel study): usage compare_libs_gemm.sh M lib...
cd "$(dirname "$0")/../.."
M=${1:-256}
shift
for rep in 1 2; do
  for lib in "$@"; do
    echo -n "m=$M $(basename $lib): "
    VLA_LIB_PATH=$PWD/$lib timeout -k 10 120 python3 tools/gemm_bench.py --m $M --iters 40 --check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:v['us_median'] for k,v in d.items() if isinstance(v,dict) and 'us_median' in v}, d.get('layer_us'))" || exit 1
  done
done
