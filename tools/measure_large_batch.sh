#!/bin/bash
# M2 at a large batch: configs[1]'s population with learn() at B = 256 (front loop), 512 and 1,024 (beyond 256 rows the
# reference's order), plain lines + per-kernel stats under rocprofv3 -> gpurun_out/<tag>/
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
TAG=${1:-r06}
O="$R/gpurun_out/$TAG"
mkdir -p "$O"
cd "$R"
: > "$O/bench_large_batch.jsonl"
for cfg in "--batch 128" "--batch 256" "--batch 256 --no-front" "--batch 512" "--batch 1024" "--batch 512 --dtype bf16" "--batch 1024 --dtype bf16"; do
  # shellcheck disable=SC2086
  python3 bench.py $cfg --no-cpu-baseline --no-sweep --steps 6000 --warmup 500 2>/dev/null | tail -1 >> "$O/bench_large_batch.jsonl"
done
for B in 512 1024; do
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_b$B" -- python3 "$R/bench.py" --no-cpu-baseline --no-sweep --steps 2000 --warmup 300 --settle-s 0.5 --batch $B > "$O/bench_batch${B}_line_under_rocprof.json" 2> "$O/bench_batch${B}_rocprof.err" )
  cp "$O"/prof_b$B/*/*kernel_stats.csv "$O/bench_batch${B}_kernel_stats.csv" 2>/dev/null
  rm -rf "$O/prof_b$B"
done
python3 - "$O/bench_large_batch.jsonl" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l)
    u = d.get("roofline_update", {})
    print(f"B={d['config']['batch']:5d} {d['dtype']:5s} loop={d['config']['loop']:16s} {d['ms_per_step']*1e3:8.2f} us/step  {d['update_steps_per_s']:9.1f} learn()/s  {d['update_samples_per_s']:12.1f} samples/s  "
          f"learn {u.get('us_per_learn')} us  frac {u.get('frac')}  value {d['value']/1e6:.1f} M env steps/s")
PY
