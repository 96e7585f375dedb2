#!/bin/bash
# kernel trace of the grouped MoE block at a decode-sized batch (kernel study): usage trace_moe.sh TOKENS
cd "$(dirname "$0")/../.."
T=${1:-256}
export TMPDIR=/tmp
rm -rf gpurun_out/moe_trace_$T
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/moe_trace_$T -o t -- python3 tools/moe_bench.py --tokens $T --grouped-only > gpurun_out/moe_trace_$T.log 2>&1 || exit 1
f=$(find gpurun_out/moe_trace_$T -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r["Name"][:110], r["Calls"], r["AverageNs"], r["Percentage"])
PY
