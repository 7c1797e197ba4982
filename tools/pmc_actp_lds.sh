#!/bin/bash
# two rocprofv3 --pmc passes (LDS activity / bank conflicts, instruction counts) over tools/pmc_actp.py; HX_LIBRARY selects the build
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
tag=${1:-x}
i=0
for set in "SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d "$R/gpurun_out/pmc_lds_${tag}_$i" -- python3 "$R/tools/pmc_actp.py" > "$R/gpurun_out/pmc_lds_${tag}_$i.log" 2>&1 || echo "pass $i failed"
done
find "$R"/gpurun_out/pmc_lds_${tag}_* -name "*.db" -delete 2>/dev/null
python3 "$R/tools/pmc_summary.py" "$R"/gpurun_out/pmc_lds_${tag}_* | cut -c1-330 | grep -v "elementwise\|copyBuffer"
