#!/bin/bash
# [r6, VERDICT r5 item 5a] launches C and D of learn() (bwd_l2<0> + wgrad<ADAM>) as ONE launch (HX_BACK_FOLD=1, hx_back.hip) against two launches: parity suites
# with the fold on, then step times alternated on one box.     tools/ubench/back_fold_ab.sh [tag]
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}
cd "$R"
TAG=${1:-r06_back_fold_ab}; O="$R/gpurun_out"; mkdir -p "$O"
{
  echo "# parity with HX_BACK_FOLD=1 (the one-call path folds; the staged / sharded sequence does not: their bit-identity tests compare the two)"
  HX_BACK_FOLD=1 timeout 900 python3 -m pytest tests/test_hirl_gpu.py tests/test_front_gpu.py tests/test_facade_gpu.py tests/test_x9_gpu.py -x -q 2>&1 | tail -4
  echo "# step times, alternated (us per step | the dominant launch's own duration)"
  for cfg in "" "--no-front" "--steps 20 --warmup 5"; do
    for rep in 1 2 3; do
      for fold in 0 1; do
        # shellcheck disable=SC2086
        HX_BACK_FOLD=$fold python3 bench.py $cfg --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print('fold=$fold', '[$cfg]', '|', round(d['ms_per_step'] * 1e3, 2), 'us/step', d['repetitions']['ms_per_step'], '| front_status', d['front_status'], '| sample+learn', d['stage_us']['sample+learn'])"
      done
    done
  done
} 2>&1 | tee "$O/$TAG.txt"
