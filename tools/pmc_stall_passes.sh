#!/bin/bash
# Stall-attribution PMC passes over tools/pmc_step.py (real training steps), one small counter group per run
# (counters only: no trace domains; every pass under its own timeout -- a group the hardware cannot schedule makes
# rocprofv3 abort and then hang).  Run on the GPU box from the repo root:  tools/pmc_stall_passes.sh [outdir] [first group]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=${1:-$R/gpurun_out/pmc_stall}
FIRST=${2:-1}
cd /tmp && export TMPDIR=/tmp
mkdir -p "$OUT"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i + 1))
  [ $i -lt $FIRST ] && continue
  rm -rf "$OUT/g$i"
  timeout 150 rocprofv3 --pmc $group -d "$OUT/g$i" --output-format csv -- python3 "$R/tools/pmc_step.py" > "$OUT/g$i.log" 2>&1 || echo "group $i failed: $group"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU
SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM
TCC_BUSY_sum TCC_CYCLE_sum TCC_TAG_STALL_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum
TCC_EA0_WRREQ_LEVEL_sum TCC_SRC_FIFO_FULL_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
GROUPS
ls "$OUT"
