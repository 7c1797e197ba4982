#!/bin/bash
# The HBM-traffic passes of the env-step kernel and of the fused act + env kernel: one counter per rocprofv3 run, program directly after `--`.
#   tools/pmc_env_passes.sh <tag>   ->  gpurun_out/<tag>/pmc_<COUNTER>_env_<n>.csv
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
O="$R/gpurun_out/${1:-r03_pmc}"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for N in 4096 65536 1048576; do
  for C in FETCH_SIZE WRITE_SIZE; do
    export HX_PMC_ENVS=$N
    if [ "$N" -le 8192 ]; then export HX_PMC_FUSED=1; else unset HX_PMC_FUSED; fi
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/p_${C}_$N" -- python3 "$R/tools/pmc_env.py" > "$O/pmc_${C}_env_$N.log" 2>&1
    cp "$O"/p_${C}_$N/*/*counter_collection.csv "$O/pmc_${C}_env_$N.csv" 2>/dev/null
    rm -rf "$O/p_${C}_$N"
  done
done
ls -la "$O"
