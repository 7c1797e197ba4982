#!/bin/bash
# [r5] The fp32 acting product as a 6-term bf16 split (-DHX_X9_TERMS=9: without lo x lo, lo x mid, mid x lo) against the exact 9-term split: accuracy against an fp64
# evaluation and step times, alternated on one box.  Build first:  tools/ubench/x9_terms_ab.sh build ;  on the GPU box: tools/ubench/x9_terms_ab.sh run [tag]
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}
cd "$R"
C=hirl4ucav_amd/csrc
if [ "${1:-run}" = build ]; then
  OBJS=""
  for f in $C/hx_*.hip; do
    b=$(basename "$f" .hip)
    if grep -q "HX_X9_TERMS\|hx_act.h\|hx_act_body.h\|hx_actp_body.h" "$f" || [ "$b" = hx_act ] || [ "$b" = hx_actp ] || [ "$b" = hx_front ]; then
      FL=""; [ "$b" = hx_env ] && FL="-ffp-contract=off"
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DHX_X9_TERMS=9 $FL -c "$f" -o "/tmp/x9f_$b.o"
      OBJS="$OBJS /tmp/x9f_$b.o"
    else
      OBJS="$OBJS $C/$b.o"
    fi
  done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o hirl4ucav_amd/libhx_x9full.so $OBJS -ldl
  ls -la hirl4ucav_amd/libhx_x9full.so
  exit 0
fi
TAG=${2:-r05_x9_terms_ab}; O="$R/gpurun_out/$TAG"; mkdir -p "$O"
{
  for L in hirl4ucav_amd/libhx_mi355.so hirl4ucav_amd/libhx_x9full.so; do
    echo "== $L: error of the acting kernel against an fp64 evaluation, 16,384 rows (max / mean), beside the fp32-MFMA kernel's"
    HX_LIBRARY="$R/$L" python3 - <<'PY'
import numpy as np, torch, sys
sys.path.insert(0, ".")
from hirl4ucav_amd.agents import engine as E
from tests import _hirl_data as D
p = D.make_params(D.PARAM_SEED)
rng = np.random.default_rng(5)
obs = rng.uniform(-1, 1, (16384, 13)).astype(np.float32)
a = {k: np.asarray(v, np.float64) for k, v in p["actor"].items()}
def ln(x, g, b):
    m = x.mean(-1, keepdims=True); v = ((x - m) ** 2).mean(-1, keepdims=True)
    return (x - m) / np.sqrt(v + 1e-5) * g + b
x = obs.astype(np.float64)
h = np.maximum(ln(x @ a["full1.weight"].T + a["full1.bias"], a["layernorm1.weight"], a["layernorm1.bias"]), 0)
h = np.maximum(ln(h @ a["full2.weight"].T + a["full2.bias"], a["layernorm2.weight"], a["layernorm2.bias"]), 0)
ref = np.tanh(h @ a["final.weight"].T + a["final.bias"])
for fmt in ("f32x9", "f32"):
    e = E.HirlEngine(batch=128)
    e.load_params(p["actor"], p["critic"], p["bc_actor"])
    e.x9_rows = None
    e.set_act_dtype(fmt)
    got = e.act(torch.from_numpy(obs).cuda()).cpu().numpy().astype(np.float64)
    d = np.abs(got - ref)
    print(f"   {fmt:6s} max {d.max():.3e} mean {d.mean():.3e}")
PY
  done
  for cfg in "" "--steps 20 --warmup 5" "--envs 65536 --scenario circular --type linear --bc_weight 0.5" "--agent sac --envs 16384 --scenario serpentine" "--envs 8192 --scenario circular --type linear --bc_weight 0.5" "--no-front --dtype f32x9"; do
    for rep in 1 2; do
      for L in hirl4ucav_amd/libhx_mi355.so hirl4ucav_amd/libhx_x9full.so; do
        # shellcheck disable=SC2086
        HX_LIBRARY="$R/$L" python3 bench.py $cfg --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$(basename "$L")', '[$cfg]', '|', round(d['value'] / 1e6, 1), 'M env steps/s', round(d['ms_per_step'] * 1e3, 2), 'us/step | dominant launch', round(d['roofline']['us_per_launch'], 2), 'us')"
      done
    done
  done
} 2>&1 | tee "$O/ab.txt"
