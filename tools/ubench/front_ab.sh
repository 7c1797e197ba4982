#!/bin/bash
# A/B on one box: the reference-order loop (--no-front) against the front loop with fp32-MFMA acting (--front-acting mfma) and with the exact-split acting format
# (the default), driver form (3 x 20) and a long run
show='import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(r["value"], r["ms_per_step"], r["repetitions"]["ms_per_step"], r["update_steps_per_s"], "launch us", r["roofline"]["us_per_launch"])'
for rep in 1 2; do
  for mode in "--no-front" "--no-front --dtype f32x9" "--front-acting mfma" ""; do
    echo "== steps 20 warmup 5 $mode"; python bench.py --steps 20 --warmup 5 --no-sweep --no-cpu-baseline $mode 2>/dev/null | python -c "$show"
    echo "== steps 2000 $mode"; python bench.py --steps 2000 --warmup 50 --reps 1 --no-sweep --no-cpu-baseline $mode 2>/dev/null | python -c "$show"
  done
done
