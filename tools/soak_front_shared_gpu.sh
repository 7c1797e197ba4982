#!/bin/bash
# The front launch's in-launch waits under CONTENTION: P processes free-run the front loop on ONE GPU at the same time (their hardware queues interleave, so a launch's
# workgroups find CUs taken by another process's), each run ending with front_check() — an in-launch wait that gave up fails its run.  The default shape's waits end
# under any dispatch order (include/hirl4ucav.h "WHY THE WAITS END").  Launch C riding (HX_FRONT_C=1: waiters can fill the chip) is NOT part of this soak: under three
# processes its waits do run into their bound and front_check() stops the run — tools/ubench/front_c_contention.sh, profiles/r05_front_c_contention.txt.
#   tools/soak_front_shared_gpu.sh [tag] [processes] [steps]   ->  gpurun_out/<tag>/soak_front_shared_gpu.txt   (about 3 minutes of GPU time)
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
O="$R/gpurun_out/${1:-soak_shared}"; P=${2:-3}; K=${3:-150000}
mkdir -p "$O"; cd "$R"
: > "$O/soak_front_shared_gpu.txt"
run_set() {  # $1 = label, $2.. = bench flags (environment: as exported by the caller)
  local label=$1; shift
  local pids=()
  for p in $(seq 1 "$P"); do
    timeout 600 python3 bench.py "$@" --steps "$K" --reps 1 --warmup 100 --no-cpu-baseline --no-sweep > "$O/p$p.out" 2> "$O/p$p.err" &
    pids+=($!)
  done
  local ok=1
  for p in $(seq 1 "$P"); do wait "${pids[$((p - 1))]}" || ok=0; done
  for p in $(seq 1 "$P"); do
    python3 - "$O/p$p.out" "$label" "$p" >> "$O/soak_front_shared_gpu.txt" <<'PY' || { ok=0; tail -3 "$O/p$p.err" >> "$O/soak_front_shared_gpu.txt"; }
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-34s process %s: %-8s steps %7d  %6.1f M env steps/s  %7.2f us per step, status word clear" % (sys.argv[2], sys.argv[3], d["config"].get("loop"), d["steps"], d["value"] / 1e6, d["ms_per_step"] * 1e3))
PY
  done
  [ "$ok" = 1 ] || echo "FAILED: $label" >> "$O/soak_front_shared_gpu.txt"
}
run_set "4,096 envs fp32 (default shape)"
run_set "4,096 envs bf16" --dtype bf16
run_set "8,192 envs fp32 (streaming role)" --envs 8192 --scenario circular --type linear --bc_weight 0.5
run_set "16,384 mixed bf16 (persistent role)" --envs 16384 --scenario mixed --dtype bf16
cat "$O/soak_front_shared_gpu.txt"
