#!/bin/bash
# default launch choice against the register-staged kernel (VLA_Q4_DMA=0) over batch sizes and
# tensor-parallel shapes (kernel study): layer totals in us
cd "$(dirname "$0")/../.."
for args in "--m 256" "--m 128" "--m 192" "--m 512" "--m 256 --tp 2" "--m 256 --tp 4" "--m 256 --tp 8" "--m 256 --model 70b --tp 8"; do
  for dma in 0 -; do
    if [ "$dma" = "-" ]; then unset VLA_Q4_DMA; else export VLA_Q4_DMA=$dma; fi
    echo -n "$args dma=$dma: "
    timeout -k 10 120 python3 tools/gemm_bench.py $args --iters 30 --check 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print({k:v['us_median'] for k,v in d.items() if isinstance(v,dict) and 'us_median' in v}, d.get('layer_us'))" || exit 1
  done
done
