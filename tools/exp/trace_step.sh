#!/bin/bash
# kernel trace of a short bench.py run, summarised per layer (tools/layer_trace.py); extra
# arguments go to bench.py; environment variables pass through.  usage: trace_step.sh <tag> [args]
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
TAG=$1; shift
OUT="$ROOT/gpurun_out/trace_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o t -- \
  python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
python3 "$ROOT/tools/layer_trace.py" "$(find $OUT -name '*kernel_trace.csv' | head -1)" > "$OUT/layers.txt"
echo "== $TAG"; cat "$OUT/layers.txt"
rm -f $(find $OUT -name '*kernel_trace.csv')
