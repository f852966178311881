#!/bin/bash
# SQ stall / activity counters of the weight-gradient kernel on the bench shape (tools/kbench_wgrad.py), one counter group per run
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_wgrad
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i + 1))
  timeout 150 rocprofv3 --pmc $group -d "$OUT/g$i" --output-format csv -- python3 "$R/tools/kbench_wgrad.py" 16 > "$OUT/g$i.log" 2>&1 || echo "group $i failed: $group"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_WAVES
SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU
SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
GRBM_GUI_ACTIVE
GROUPS
python3 "$R/tools/pmc_stall_summary.py" "$OUT" "$OUT/wgrad_pmc.csv"
rm -rf "$OUT"/g*/
