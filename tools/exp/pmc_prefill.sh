#!/bin/bash
# counter passes over the prompt attention kernel (3 launches of 8 x 4096 tokens, Llama-3-8B heads);
# usage: pmc_prefill.sh [nseq seqlen]; extra environment (VLA_PF_FORM=1 ...) is passed through
ROOT="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}"
NS=${1:-8}
SL=${2:-4096}
OUT="$ROOT/gpurun_out/pmc_prefill_${NS}x${SL}${TAG}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for set in \
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
  "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
  "SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES" \
  "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CU_CYCLES" \
  "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" ; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $set --output-format csv -d "$OUT/p$i" -o g -- \
    python3 "$ROOT/tools/exp/prefill_stamp.py" $NS $SL > "$OUT/p$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$OUT/p$i.log"; }
  f=$(find "$OUT/p$i" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 "$ROOT/tools/profile_summary.py" pmc "$f" "$OUT/p$i.csv" && grep "prefill" "$OUT/p$i.csv" | cut -d, -f1-4 | sed 's/^_ZN3vla12_GLOBAL__N_1..//' | cut -c1-140
done
