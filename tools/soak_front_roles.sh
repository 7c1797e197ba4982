#!/bin/bash
# Long free-running soaks of every acting role of the front launch (bench.py ends with front_check(): an in-launch wait that gave up fails the run).
#   tools/soak_front_roles.sh [tag]   ->  gpurun_out/<tag>/soak_front_roles.jsonl   (about 3 minutes of GPU time)
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
O="$R/gpurun_out/${1:-soak}"
mkdir -p "$O"
cd "$R"
: > "$O/soak_front_roles.jsonl"
while read -r cfg; do
  [ -z "$cfg" ] && continue
  # shellcheck disable=SC2086
  timeout 300 python3 bench.py $cfg --reps 1 --warmup 100 --no-cpu-baseline --no-sweep 2>"$O/soak.err" | tail -1 >> "$O/soak_front_roles.jsonl" || { echo "FAILED: $cfg"; tail -5 "$O/soak.err"; }
done <<'CFGS'
--dtype bf16 --steps 400000
--envs 16384 --scenario mixed --dtype bf16 --steps 300000
--envs 8192 --scenario circular --type linear --bc_weight 0.5 --steps 300000
--envs 16384 --scenario mixed --steps 200000
--agent sac --envs 16384 --scenario serpentine --steps 200000
--staged --steps 300000
--envs 40000 --scenario mixed --dtype bf16 --steps 100000
--envs 20000 --scenario mixed --steps 100000
CFGS
python3 - "$O/soak_front_roles.jsonl" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    print("%-110s %-16s steps %7d  %7.1f M env steps/s  %7.2f us" % (d["config"]["workload"][:110], d["config"].get("loop"), d["steps"], d["value"] / 1e6, d["ms_per_step"] * 1e3))
PY
