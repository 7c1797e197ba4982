"""Gaps between consecutive kernels of the timed loop, from a rocprofv3 --kernel-trace CSV: for every (previous kernel, next kernel) pair the median of
next.start - previous.end and of the kernels' durations.
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gaps -- python3 bench.py --no-cpu-baseline --no-sweep --steps 3000 --warmup 300 --reps 1)
    python3 tools/ubench/kernel_gaps.py gpurun_out/gaps/*/*kernel_trace.csv"""
import csv
import re
import sys
from collections import defaultdict

import numpy as np


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"([A-Za-z0-9_]+)(<[^>]*>)?", n)
    return (m.group(1) + (m.group(2) or ""))[:44] if m else n[:44]


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
rows.sort()
rows = rows[len(rows) // 3:]  # the timed region (the set-up and the warm-up come first)
gaps, durs = defaultdict(list), defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gaps[n0, n1].append(s1 - e0)
    durs[n1].append(e1 - s1)
print("%-46s -> %-46s %7s %9s %9s" % ("previous kernel", "next kernel", "count", "gap us", "next us"))
for (a, b), v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) < 50:
        continue
    print("%-46s -> %-46s %7d %9.2f %9.2f" % (a, b, len(v), np.median(v) / 1e3, np.median(durs[b]) / 1e3))
