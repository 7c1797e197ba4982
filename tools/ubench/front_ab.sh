#!/bin/bash
# A/B on one box: the default loop against --front, driver form (3 x 20) and a long run
for rep in 1 2; do
  for mode in "" "--front"; do
    echo "== steps 20 warmup 5 $mode"; python bench.py --steps 20 --warmup 5 --no-sweep --no-cpu-baseline $mode | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['repetitions']['ms_per_step'], r['stage_us'], r['roofline']['us_per_launch'])"
    echo "== steps 2000 $mode"; python bench.py --steps 2000 --warmup 50 --reps 1 --no-sweep --no-cpu-baseline $mode | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r['value'], r['ms_per_step'], r['update_steps_per_s'])"
  done
done
