#!/bin/bash
# counter passes (8 SQ counters each) over one grouped int4 launch; usage: pmc_grouped.sh <w13|w2> <tokens>
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
WHICH=${1:-w13}
TOK=${2:-256}
OUT="$ROOT/gpurun_out/pmc_grouped_${WHICH}_$TOK"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
  "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU" \
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH SQ_INSTS_SMEM SQ_WAVES" \
  "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum" ; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -o g -- \
    python3 "$ROOT/tools/exp/grouped_probe.py" --which $WHICH --tokens $TOK --iters 4 > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/p$i.log"; }
  f=$(find "$OUT/p$i" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 "$ROOT/tools/profile_summary.py" pmc "$f" "$OUT/p$i.csv" && grep "q4_gemm" "$OUT/p$i.csv" | cut -d, -f1-4 | sed 's/^_ZN3vla12_GLOBAL__N_1..//' | cut -c1-120
done
