#!/bin/bash
# A/B of launch C (the critics' backward) riding inside the front launch (HX_FRONT_C=1) against a launch of its own (0), several populations
#   tools/ubench/front_c_ab.sh
cd "${GRAFT_REPO_ROOT:-.}"
while read -r cfg; do
  [ -z "$cfg" ] && continue
  for c in 1 0 1 0; do
    # shellcheck disable=SC2086
    HX_FRONT_C=$c python3 bench.py $cfg --no-cpu-baseline --no-sweep --reps 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('HX_FRONT_C=$c', '$cfg', '|', round(d['value']/1e6,1), 'M', round(d['ms_per_step']*1e3,2), 'us', d['update_steps_per_s'])"
  done
done <<'CFGS'
--envs 131072 --scenario mixed --dtype bf16 --steps 4000 --warmup 500
--envs 65536 --scenario circular --type linear --bc_weight 0.5 --steps 4000 --warmup 500
--envs 16384 --scenario mixed --dtype bf16
--envs 8192 --scenario circular --type linear --bc_weight 0.5
--dtype bf16
CFGS
