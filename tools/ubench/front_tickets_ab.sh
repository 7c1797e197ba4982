#!/bin/bash
# Roles of the front launch's update workgroups by START order (a ticket) against by index: apply tools/ubench/front_tickets_prototype.patch, make, run this
# on the GPU box, revert (profiles/r05_front_tickets_ab.txt: +2.8 us per step, not kept).
cd $GRAFT_REPO_ROOT
HX_FRONT_TICKETS=1 python -m pytest tests/test_front_gpu.py -m gpu -x -q 2>&1 | tail -3
for rep in 1 2 3; do
 for t in 0 1; do
  HX_FRONT_TICKETS=$t python bench.py --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tickets=$t default', round(d['ms_per_step']*1e3,2), 'us', 'front launch', d['roofline']['us_per_launch'])"
  HX_FRONT_TICKETS=$t python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sweep 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tickets=$t driver form', round(d['ms_per_step']*1e3,2), 'us', [round(x*1e3,1) for x in d['repetitions']['ms_per_step']])"
 done
done
