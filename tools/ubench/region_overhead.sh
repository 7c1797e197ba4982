#!/bin/bash
# What does a timed region of K steps cost beyond K x (steady-state step)?  bench.py at K = 20 / 80 / 320 / 5000; the fixed part is
# (ms_per_step(K) - ms_per_step(5000)) x K.  (Round 3: 20-30 us per region = +1.5 us per step in the driver's 20-step form; polling the stream
# before torch.cuda.synchronize() changed nothing — it is the idle GPU waiting for the host's first launches, not the wake-up at the end.)
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/../.." && pwd)}
cd "$R"
for poll in ""; do
  for K in 20 80 320 5000; do
    python3 bench.py --steps $K --warmup 5 --no-cpu-baseline --no-sweep $poll "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('poll' if '$poll' == '' else 'sync', 'K=%5d' % $K, 'us/step', [round(1e3 * x, 2) for x in d['repetitions']['ms_per_step']], 'median', round(1e3 * d['ms_per_step'], 2))"
  done
done
