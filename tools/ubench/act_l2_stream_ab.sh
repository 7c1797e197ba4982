#!/bin/bash
# [r6, VERDICT r5 item 4] What would the per-tile acting product gain if the W2-image stream out of L2 cost NOTHING — the ceiling of every scheme that removes it
# (column-stationary workgroups with the image slab in LDS, ...)?  Two timing-only builds (-DHX_DBG_ACT_HOT=1 / =2, hx_act_body.h: the same load instructions and
# bytes per lane, from 6 KB per wave / per workgroup instead of the 768 KB image; results are wrong by construction) against the product library, alternated on
# one box: the front launch's own duration (live stamps) and the step.     build first:  tools/ubench/act_l2_stream_ab.sh build ;  GPU box: ... run [tag]
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}
cd "$R"
C=hirl4ucav_amd/csrc
if [ "${1:-run}" = build ]; then
  for lvl in 1 2; do
    OBJS=""
    for f in $C/hx_*.hip; do
      b=$(basename "$f" .hip)
      if [ "$b" = hx_act ] || [ "$b" = hx_front ]; then
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DHX_DBG_ACT_HOT=$lvl -c "$f" -o "/tmp/hot${lvl}_$b.o" &
        OBJS="$OBJS /tmp/hot${lvl}_$b.o"
      else
        OBJS="$OBJS $C/$b.o"
      fi
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o hirl4ucav_amd/libhx_hot$lvl.so $OBJS -ldl
  done
  ls -la hirl4ucav_amd/libhx_hot*.so
  exit 0
fi
TAG=${2:-r06_act_l2_stream_ab}; O="$R/gpurun_out"; mkdir -p "$O"
{
  echo "# product library | HX_DBG_ACT_HOT=1 (B fragments from 6 KB per wave: L2-hot) | =2 (6 KB per workgroup: L1-hot); timing only, alternated on one box"
  for cfg in "" "--no-front" "--envs 8192 --scenario circular --type linear --bc_weight 0.5 --no-front"; do
    for rep in 1 2 3; do
      for L in libhx_mi355.so libhx_hot1.so libhx_hot2.so; do
        # shellcheck disable=SC2086
        HX_LIBRARY="$R/hirl4ucav_amd/$L" python3 bench.py $cfg --no-cpu-baseline --no-sweep --steps 6000 --warmup 500 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$L', '[$cfg]', '|', round(d['ms_per_step'] * 1e3, 2), 'us/step | dominant launch', round(d['roofline']['us_per_launch'], 2), 'us | act alone', d.get('roofline_act', {}).get('us'))"
      done
    done
  done
} 2>&1 | tee "$O/$TAG.txt"
