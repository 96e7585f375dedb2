for sp in 1 2 4 8 16; do
  echo "splits=$sp"
  VLA_Q4_TUNE="4,$sp" timeout -k 10 120 python tools/gemm_bench.py --m 16 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:(v['us_median'] if isinstance(v,dict) else round(v,1)) for k,v in d.items()})" || exit 1
done
