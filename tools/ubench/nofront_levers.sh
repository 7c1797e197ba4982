#!/bin/bash
# [r5] VERDICT r4 item 5: what the front launch taught, tried on the REFERENCE-ORDER loop (bench.py --no-front): launch B in 64-column workgroups (HX_FWD_NT64=1)
# and the exact-split acting format at 4,096 envs (--dtype f32x9), alternated on one box -> gpurun_out/<tag>/ab.txt
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}
TAG=${1:-r05_nofront_levers}; O="$R/gpurun_out/$TAG"; mkdir -p "$O"; cd "$R"
line() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$1', round(d['ms_per_step'] * 1e3, 2), 'us/step', round(d['value'] / 1e6, 2), 'M env steps/s', round(d.get('update_steps_per_s', 0)), 'learn()/s')"; }
{
  for rep in 1 2 3; do
    python3 bench.py --no-front --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 | line "reference order, default             "
    HX_FWD_NT64=1 python3 bench.py --no-front --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 | line "reference order, launch B in 64 columns"
    python3 bench.py --no-front --dtype f32x9 --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 | line "reference order, exact-split acting   "
    python3 bench.py --no-front --steps 20 --warmup 5 --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 | line "reference order, default, driver form "
    python3 bench.py --no-front --dtype f32x9 --steps 20 --warmup 5 --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 | line "reference order, exact-split, driver form"
  done
} 2>&1 | tee "$O/ab.txt"
