#!/bin/bash
# The step's launches on the device clock: rocprofv3 --kernel-trace of tools/pmc_learn.py (the bench loop, default configuration, no events or stamps) ->
# per kernel the mean duration, the mean idle gap in front of it (previous kernel's end -> this kernel's begin) and, from two separate --pmc passes,
# FETCH_SIZE / WRITE_SIZE per launch.  -> gpurun_out/<tag>_step_timeline.txt       tools/step_timeline.sh <tag> [bench flags]
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
TAG=${1:-r06}; shift || true
O="$R/gpurun_out/${TAG}_timeline"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
HX_PMC_STEPS=400 rocprofv3 --kernel-trace --output-format csv -d "$O/trace" -- python3 "$R/tools/pmc_learn.py" "$@" > "$O/trace.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  HX_PMC_STEPS=100 rocprofv3 --pmc $c --output-format csv -d "$O/pmc_$c" -- python3 "$R/tools/pmc_learn.py" "$@" > "$O/pmc_$c.log" 2>&1 || echo "pass $c failed"
done
python3 - "$O" <<'PY' | tee "$R/gpurun_out/${TAG}_step_timeline.txt"
import csv, glob, sys, collections, re
O = sys.argv[1]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:70]
rows = []
for f in glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
rows = rows[len(rows) // 3:]  # steady state
dur, gap, prev = collections.defaultdict(list), collections.defaultdict(list), collections.defaultdict(collections.Counter)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    dur[n1].append(e1 - s1); gap[n1].append(s1 - e0); prev[n1][n0] += 1
pmc = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(O + f"/pmc_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    pmc[c] = {k: sum(v) / len(v) for k, v in acc.items()}
print("# rocprofv3 --kernel-trace: per kernel [begin, end] on the device clock; gap = previous kernel's end -> this kernel's begin (same queue)")
print("# FETCH_SIZE / WRITE_SIZE: separate --pmc passes, raw counter units (KB on gfx950: FETCH in 64-B x 2 per the guide's calibration -> tools/pmc_traffic_json.py)")
print(f"{'kernel':72s} {'calls':>6s} {'dur us':>8s} {'gap us':>8s} {'FETCH':>10s} {'WRITE':>10s}  mostly after")
tot = 0.0
for n in sorted(dur, key=lambda k: -sum(dur[k])):
    d, g = sum(dur[n]) / len(dur[n]) / 1e3, sum(gap[n]) / len(gap[n]) / 1e3
    print(f"{n:72s} {len(dur[n]):6d} {d:8.2f} {g:8.2f} {pmc['FETCH_SIZE'].get(n, float('nan')):10.1f} {pmc['WRITE_SIZE'].get(n, float('nan')):10.1f}  {prev[n].most_common(1)[0][0][:40]}")
span = (rows[-1][1] - rows[0][0]) / 1e3
busy = sum(e - s for s, e, _ in rows) / 1e3
print(f"# window {span:.0f} us, kernels busy {busy:.0f} us = {busy / span:.3f}; idle between kernels {span - busy:.0f} us")
PY
