#!/bin/bash
# The HBM-traffic passes of the env-step kernel and of the fused act + env kernels: one counter per rocprofv3 run, program directly after `--`.
#   tools/pmc_env_passes.sh <tag>   ->  gpurun_out/<tag>/pmc_<COUNTER>_env_<n>[_<dtype>].csv
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
O="$R/gpurun_out/${1:-r04_pmc}"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# envs per launch : fused act + env launches too (0 / 1) : policy format of the fused launches
for CFG in 4096:2:f32 8192:2:f32 65536:1:f32 1048576:0:f32 16384:2:bf16 131072:1:bf16; do
  IFS=: read -r N FUSED DT <<< "$CFG"
  for C in FETCH_SIZE WRITE_SIZE; do
    export HX_PMC_ENVS=$N HX_PMC_DTYPE=$DT
    if [ "$FUSED" -ge 1 ]; then export HX_PMC_FUSED=1; else unset HX_PMC_FUSED; fi
    if [ "$FUSED" = 2 ]; then export HX_PMC_FRONT=1; else unset HX_PMC_FRONT; fi   # 2: + the front launch (bench.py's default loop at this size)
    SUF=$([ "$DT" = f32 ] && echo "" || echo "_$DT")
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/p_${C}_$N$SUF" -- python3 "$R/tools/pmc_env.py" > "$O/pmc_${C}_env_$N$SUF.log" 2>&1
    cp "$O"/p_${C}_$N$SUF/*/*counter_collection.csv "$O/pmc_${C}_env_$N$SUF.csv" 2>/dev/null
    rm -rf "$O/p_${C}_$N$SUF"
  done
done
ls -la "$O"
