#!/bin/bash
# Where a small-batch decode step's time goes: the unprofiled step time beside the kernel trace of
# the same command (sum of kernel durations per step = what is left after launch gaps).
#   bash tools/exp/small_batch_step.sh <batch> [extra bench.py flags]
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
B=$1; shift
OUT="$ROOT/gpurun_out/small_b$B"
mkdir -p "$OUT"
cd "$ROOT"
timeout -k 10 200 python3 bench.py --batch $B --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- \
  python3 "$ROOT/bench.py" --batch $B --steps 20 --warmup 3 --no-cpu-baseline "$@" > "$OUT/bench_profiled.json" 2> "$OUT/bench_profiled.err" || exit 2
cd "$ROOT"
python3 tools/profile_summary.py stats "$(find $OUT/trace -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_stats.csv" 30
rm -rf "$OUT/trace"
