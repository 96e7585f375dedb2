#!/bin/bash
# Collect the round's profiles on an MI355X box: kernel trace of bench.py, PMC passes on the int4
# GEMM (one counter family per pass), the unprofiled bench line.  Results go to gpurun_out/prof/;
# the summaries that are kept are copied into profiles/ afterwards (tools/profile_summary.py).
set -o pipefail
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$ROOT/gpurun_out/prof"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/bench" -o bench -- \
  python3 "$ROOT/bench.py" --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_profiled.json" 2> "$OUT/bench_profiled.err" || exit 1
# HBM traffic of the paged-attention kernel the benchmark launches (metric shape, unique blocks)
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/attn_fetch" -o a -- \
  python3 "$ROOT/tools/attn_bench.py" --iters 3 > /dev/null 2>&1 || exit 6
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/attn_write" -o a -- \
  python3 "$ROOT/tools/attn_bench.py" --iters 3 > /dev/null 2>&1 || exit 7
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/gemm_fetch" -o g -- \
  python3 "$ROOT/tools/gemm_bench.py" --m 256 --iters 4 --sets 4 > /dev/null 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/gemm_write" -o g -- \
  python3 "$ROOT/tools/gemm_bench.py" --m 256 --iters 4 --sets 4 > /dev/null 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE \
  --output-format csv -d "$OUT/gemm_sq" -o g -- \
  python3 "$ROOT/tools/gemm_bench.py" --m 256 --iters 4 --sets 4 > /dev/null 2>&1 || exit 4
cd "$ROOT"
timeout -k 10 300 python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 5
S=tools/profile_summary.py
python3 $S stats "$(find $OUT/bench -name '*kernel_stats.csv' | head -1)" "$OUT/bench_kernel_stats.csv" 40
python3 $S record "$(find $OUT/bench -name '*kernel_stats.csv' | head -1)" "$OUT/kernel_times.json" \
  "profiles/${ROUND:-r03}_bench_kernel_stats.csv: rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"
python3 $S pmc "$(find $OUT/attn_fetch -name '*counter_collection.csv' | head -1)" "$OUT/attn_pmc_fetch.csv"
python3 $S pmc "$(find $OUT/attn_write -name '*counter_collection.csv' | head -1)" "$OUT/attn_pmc_write.csv"
python3 $S attn_traffic "$OUT/attn_pmc_fetch.csv" "$OUT/attn_pmc_write.csv" "$OUT/attn_traffic.json" \
  "rocprofv3 --pmc FETCH_SIZE -- python3 tools/attn_bench.py --iters 3 (and the same with WRITE_SIZE)"
python3 $S pmc "$(find $OUT/gemm_fetch -name '*counter_collection.csv' | head -1)" "$OUT/gemm_pmc_fetch.csv"
python3 $S pmc "$(find $OUT/gemm_write -name '*counter_collection.csv' | head -1)" "$OUT/gemm_pmc_write.csv"
python3 $S pmc "$(find $OUT/gemm_sq -name '*counter_collection.csv' | head -1)" "$OUT/gemm_pmc.csv"
python3 $S gemm_traffic "$OUT/gemm_pmc_fetch.csv" "$OUT/gemm_pmc_write.csv" "$OUT/gemm_traffic.json" \
  "rocprofv3 --pmc FETCH_SIZE -- python3 tools/gemm_bench.py --m 256 --iters 4 --sets 4 (and the same with WRITE_SIZE)"
cat "$OUT/bench.json"
