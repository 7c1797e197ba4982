#!/bin/bash
# rocprofv3 --pmc passes (SQ counters) over tools/pmc_actp.py -> gpurun_out/pmc_actp_[1-6]; summary: python tools/pmc_summary.py gpurun_out/pmc_actp_*
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
mkdir -p "$R/gpurun_out"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_IFETCH" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT" "SQ_WAVES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$R/gpurun_out/pmc_actp_$i" -- python3 "$R/tools/pmc_actp.py" > "$R/gpurun_out/pmc_actp_$i.log" 2>&1 || echo "pass $i failed"
  tail -2 "$R/gpurun_out/pmc_actp_$i.log" | cut -c1-200
done
find "$R"/gpurun_out/pmc_actp_* -name "*.db" -delete 2>/dev/null
du -sh "$R"/gpurun_out/pmc_actp_*
