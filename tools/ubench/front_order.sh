#!/bin/bash
# the dispatch order of the update's jobs inside the front launch (HX_FRONT_ORDER: 0 = launch A then launch B (default), 1 = target critics right
# behind the target actor and Q1, 2 = the critic call's jobs then the actor call's extras), 2,000-step runs alternated on one box: us per step, the front launch's own duration
for rep in 1 2 3; do for v in 0 1 2; do
  HX_FRONT_ORDER=$v python bench.py --steps 2000 --warmup 100 --reps 1 --no-sweep --no-cpu-baseline 2>/dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('order $v', round(r['ms_per_step']*1e3,2), 'us/step, front launch', r['roofline']['us_per_launch'])"
done; done
