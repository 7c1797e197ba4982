#!/bin/bash
# [r5] How many CUs the persistent bf16 acting role of the front launch takes at 16,384 envs (HX_FRONT_PERSIST_WGS: 176 = two thirds, the default) and the order of the
# update's jobs at 4,096 envs (HX_FRONT_ORDER), alternated on one box -> gpurun_out/<tag>/ab.txt
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}
TAG=${1:-r05_front_persist_wgs}; O="$R/gpurun_out/$TAG"; mkdir -p "$O"; cd "$R"
line() { python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$1', round(d['ms_per_step'] * 1e3, 2), 'us/step', round(d['value'] / 1e6, 2), 'M env steps/s | front launch', round(d['roofline']['us_per_launch'], 2), 'us')"; }
{
  for rep in 1 2; do
    for w in 176 128 144 160 192; do
      HX_FRONT_PERSIST_WGS=$w python3 bench.py --envs 16384 --scenario mixed --dtype bf16 --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 | line "16,384 mixed bf16, acting workgroups <= $w"
    done
  done
  for rep in 1 2; do
    for w in 176 128 256; do
      HX_FRONT_PERSIST_WGS=$w python3 bench.py --envs 8192 --scenario mixed --dtype bf16 --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 | line "8,192 mixed bf16, acting workgroups <= $w"
    done
  done
  for rep in 1 2; do
    for o in 0 1 2; do
      HX_FRONT_ORDER=$o python3 bench.py --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 | line "4,096 fp32, HX_FRONT_ORDER=$o"
    done
  done
} 2>&1 | tee "$O/ab.txt"
