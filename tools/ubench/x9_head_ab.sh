#!/bin/bash
# [r5] The exact-split (fp32) STREAMING acting kernel with LayerNorm 2 + the final layer straight from the accumulators (an exact variant of the bf16 kernels' head:
# h2 and W3 as hi | mid | lo parts, six partial products) against the z2-tile head: error against fp64 and step times, two builds alternated on one box.
#   tools/ubench/x9_head_ab.sh [libA] [libB]     (defaults: the product library, hirl4ucav_amd/libhx_A.so)
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}
cd "$R"
LA=${1:-hirl4ucav_amd/libhx_mi355.so}; LB=${2:-hirl4ucav_amd/libhx_A.so}
for L in $LA $LB; do
  echo "== $L: error of the acting kernels against an fp64 evaluation (max / mean): 16,384 rows (streaming kernel) and 4,096 rows (per-tile kernel), HIRL policy"
  HX_LIBRARY="$R/$L" python3 - <<'PY'
import numpy as np, torch, sys
sys.path.insert(0, ".")
from hirl4ucav_amd.agents import engine as E
from tests import _hirl_data as D
p = D.make_params(D.PARAM_SEED)
a = {k: np.asarray(v, np.float64) for k, v in p["actor"].items()}
def ln(x, g, b):
    m = x.mean(-1, keepdims=True); v = ((x - m) ** 2).mean(-1, keepdims=True)
    return (x - m) / np.sqrt(v + 1e-5) * g + b
for n in (16384, 4096):
    rng = np.random.default_rng(5)
    obs = rng.uniform(-1, 1, (n, 13)).astype(np.float32)
    x = obs.astype(np.float64)
    h = np.maximum(ln(x @ a["full1.weight"].T + a["full1.bias"], a["layernorm1.weight"], a["layernorm1.bias"]), 0)
    h = np.maximum(ln(h @ a["full2.weight"].T + a["full2.bias"], a["layernorm2.weight"], a["layernorm2.bias"]), 0)
    ref = np.tanh(h @ a["final.weight"].T + a["final.bias"])
    e = E.HirlEngine(batch=128)
    e.load_params(p["actor"], p["critic"], p["bc_actor"])
    e.x9_rows = None
    e.set_act_dtype("f32x9")
    got = e.act(torch.from_numpy(obs).cuda()).cpu().numpy().astype(np.float64)
    d = np.abs(got - ref)
    print(f"   n {n:6d}  max {d.max():.3e} mean {d.mean():.3e}")
PY
done
for cfg in "--envs 65536 --scenario circular --type linear --bc_weight 0.5" "--agent sac --envs 16384 --scenario serpentine" "--envs 8192 --scenario circular --type linear --bc_weight 0.5" "--envs 16384 --scenario mixed" ""; do
  for rep in 1 2; do
    for L in $LA $LB; do
      # shellcheck disable=SC2086
      HX_LIBRARY="$R/$L" python3 bench.py $cfg --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 | python3 -c "
import json, sys
d = json.loads(sys.stdin.read())
print('$(basename "$L")', '[$cfg]', '|', round(d['value'] / 1e6, 1), 'M env steps/s', round(d['ms_per_step'] * 1e3, 2), 'us/step | dominant launch', round(d['roofline']['us_per_launch'], 2), 'us')"
    done
  done
done
