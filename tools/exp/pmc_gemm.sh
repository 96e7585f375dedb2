#!/bin/bash
# counter passes (8 SQ counters each) over the int4 GEMM of one layer shape; usage: pmc_gemm.sh <shape> <VLA_Q4_DMA>
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
ONLY=${1:-gate_up}
export VLA_Q4_DMA=${2:-0}
OUT="$ROOT/gpurun_out/pmc_$ONLY_$VLA_Q4_DMA"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
  "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH SQ_INSTS_SMEM SQ_WAVES" ; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -o g -- \
    python3 "$ROOT/tools/gemm_bench.py" --m 256 --iters 4 --sets 4 --only $ONLY > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/p$i.log"; }
  f=$(find "$OUT/p$i" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 "$ROOT/tools/profile_summary.py" pmc "$f" "$OUT/p$i.csv" && grep "q4_gemm" "$OUT/p$i.csv" | cut -d, -f1-4 | sed 's/^_ZN3vla12_GLOBAL__N_118//' | cut -c1-120
done
