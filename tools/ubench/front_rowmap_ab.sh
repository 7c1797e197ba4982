#!/bin/bash
# [r5] Which of the front launch's update workgroups share an XCD: a row tile's (HX_FRONT_ROWMAP=1, default: the later launches find their rows in their own L2)
# or a 64-column slice's (HX_FRONT_ROWMAP=0: every W2 slice of launches A / B enters ONE L2 instead of eight).  Step time alternated on one box (driver
# form and 3 x 20,000 steps) and the front launch's HBM traffic from --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs) -> gpurun_out/<tag>/
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}
TAG=${1:-r05_front_rowmap_ab}; O="$R/gpurun_out/$TAG"; mkdir -p "$O"; cd "$R"
line() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$1', round(d['ms_per_step'] * 1e3, 2), 'us/step', round(d['value'] / 1e6, 2), 'M env steps/s | front launch', round(d['roofline']['us_per_launch'], 2), 'us')"; }
{
  for rep in 1 2 3; do
    for m in 1 0; do
      HX_FRONT_ROWMAP=$m python3 bench.py --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 | line "rowmap=$m default-run"
      HX_FRONT_ROWMAP=$m python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 | line "rowmap=$m driver-form"
    done
  done
  for m in 1 0; do
    HX_FRONT_ROWMAP=$m python3 bench.py --envs 8192 --scenario circular --type linear --bc_weight 0.5 --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 | line "rowmap=$m circ8192"
    HX_FRONT_ROWMAP=$m python3 bench.py --dtype bf16 --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 | line "rowmap=$m bf16-4096"
  done
} | tee "$O/ab.txt"
[ -n "${NO_PMC:-}" ] && exit 0
cd /tmp && export TMPDIR=/tmp
for m in 1 0; do
  for C in FETCH_SIZE WRITE_SIZE; do
    export HX_FRONT_ROWMAP=$m HX_PMC_ENVS=4096 HX_PMC_DTYPE=f32 HX_PMC_FUSED=1 HX_PMC_FRONT=1
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$O/p_${C}_$m" -- python3 "$R/tools/pmc_env.py" > "$O/pmc_${C}_rowmap$m.log" 2>&1
    mkdir -p "$O/rowmap$m"
    cp "$O"/p_${C}_$m/*/*counter_collection.csv "$O/rowmap$m/pmc_${C}_env_4096.csv" 2>/dev/null
    rm -rf "$O/p_${C}_$m"
  done
  python3 "$R/tools/pmc_traffic_json.py" "$O/rowmap$m" "$(cd "$R" && git rev-parse --short HEAD 2>/dev/null || echo worktree)" > "$O/traffic_rowmap$m.json"
  python3 -c "
import json; d = json.load(open('$O/traffic_rowmap$m.json'))['front_4096']; print('rowmap=$m front_4096:', d['traffic_bytes'], 'B per launch =', d['ratio'], 'x the algorithmic', d['algorithmic_bytes'])" | tee -a "$O/ab.txt"
done
