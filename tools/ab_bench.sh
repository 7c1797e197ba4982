#!/bin/bash
# A/B timing of two builds of libhx_mi355.so on ONE box: alternate `bench.py --no-sweep --no-cpu-baseline` runs, print us/step of each.
#   tools/ab_bench.sh hirl4ucav_amd/libhx_mi355_A.so hirl4ucav_amd/libhx_mi355.so [rounds] [bench flags...]
set -euo pipefail
A=$(realpath "$1"); B=$(realpath "$2"); N=${3:-3}; shift 3 || shift $#
cd "${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}"
for i in $(seq "$N"); do
  for L in "$A" "$B"; do
    HX_LIBRARY="$L" python3 bench.py --no-sweep --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$(basename "$L")', round(d['ms_per_step']*1e3,2), 'us/step', d.get('update_steps_per_s'))"
  done
done
