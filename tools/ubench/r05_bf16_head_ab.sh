#!/bin/bash
# [r5] The bf16 acting kernels with LayerNorm 2 + the final layer straight from the accumulators (hx_act.h) against the build before (z2 tile in LDS, 16-lane
# head): alternated on ONE box.   tools/ubench/r05_bf16_head_ab.sh <old.so> <new.so> [tag]   -> gpurun_out/<tag>/
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}
A=$(realpath "$1"); B=$(realpath "$2"); TAG=${3:-r05_bf16_head_ab}
O="$R/gpurun_out/$TAG"; mkdir -p "$O"; cd "$R"
./tools/ubench/permlane_probe.bin | tee "$O/permlane_probe.txt"
{
  for rep in 1 2; do
    for L in "$A" "$B"; do
      echo "== $(basename "$L") (pass $rep): acting launch alone / act + env + insert, bf16"
      HX_LIBRARY="$L" SIZES=4096,8192,16384,65536,131072 python3 tools/ubench/actp_time.py bf16 2>/dev/null
    done
  done
  for cfg in "--envs 16384 --scenario mixed --dtype bf16" "--envs 131072 --scenario mixed --dtype bf16" "--dtype bf16" "--envs 16384 --scenario mixed --dtype bf16 --no-front"; do
    for rep in 1 2; do
      for L in "$A" "$B"; do
        # shellcheck disable=SC2086
        HX_LIBRARY="$L" python3 bench.py $cfg --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
r = d['roofline']
print('$(basename "$L")', '$cfg', '|', round(d['value'] / 1e6, 1), 'M env steps/s', round(d['ms_per_step'] * 1e3, 2), 'us/step |', r['kernel'][:40], round(r['us_per_launch'], 2), 'us')"
      done
    done
  done
} | tee "$O/ab.txt"
