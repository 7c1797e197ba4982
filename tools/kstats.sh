#!/bin/bash
# per-kernel mean durations of a short bench run under rocprofv3 --kernel-trace --stats (csv), top kernels only
#   tools/kstats.sh [label] [bench flags...]      (HX_LIBRARY selects the build)
set -euo pipefail
L=${1:-k}; shift || true
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/kstats_$L" -- python3 "$R/bench.py" --no-sweep --no-cpu-baseline --steps 4000 --warmup 500 --settle-s 0.5 "$@" > "$R/gpurun_out/kstats_$L.log" 2>&1
f=$(ls "$R"/gpurun_out/kstats_"$L"/*/*kernel_stats.csv | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    n=r["Name"].replace("(anonymous namespace)::","").split("(")[0].replace("void ","")
    print("%-46s calls %6s  mean %8.2f us  %5s%%" % (n[:46], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"][:5]))
PY
grep "^{" "$R/gpurun_out/kstats_$L.log" | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(\"under rocprof:\", round(d[\"ms_per_step\"]*1e3,2), \"us/step\")"
