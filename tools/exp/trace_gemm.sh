#!/bin/bash
# kernel trace of tools/gemm_bench.py at batch M: average duration per kernel (kernel study)
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
M=${1:-16}
OUT="$ROOT/gpurun_out/trace_gemm_m$M"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for only in qkv o gate_up down; do
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$only" -o t -- \
    python3 "$ROOT/tools/gemm_bench.py" --m $M --iters 20 --only $only > "$OUT/$only.log" 2>&1 || exit 1
  echo "== $only: $(tail -1 $OUT/$only.log | cut -c1-120)"
  python3 - "$(find $OUT/$only -name '*kernel_stats.csv' | head -1)" <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'q4_' in r['Name']:
        print("   ", r['Name'][23:75], r['Calls'], round(float(r['AverageNs'])/1e3,2), 'us')
PY
done
