#!/bin/bash
# Launch C riding in the front launch (HX_FRONT_C=1: waiting workgroups can fill the chip) with 1, 2, 3 processes free-running on ONE GPU: does a wait run into its bound
# when another process's workgroups take the CUs?  (profiles/r05_front_c_contention.txt)
cd $GRAFT_REPO_ROOT
for P in 1 2 3; do
  echo "== $P process(es), launch C riding (HX_FRONT_C=1), 8,192 envs, 3,000 steps each"
  pids=()
  for p in $(seq 1 $P); do
    ( timeout 200 env HX_FRONT_C=1 python3 bench.py --envs 8192 --scenario circular --type linear --bc_weight 0.5 --steps 3000 --reps 1 --warmup 100 --no-cpu-baseline --no-sweep > /tmp/cp$p.out 2> /tmp/cp$p.err; echo "exit code $?" >> /tmp/cp$p.err ) &
    pids+=($!)
  done
  wait
  for p in $(seq 1 $P); do
    python3 -c "
import json,sys
t=open('/tmp/cp$p.out').read().strip().splitlines()
print('  process $p:', ('%.2f us per step' % (json.loads(t[-1])['ms_per_step']*1e3)) if t else 'no line')"
    grep -v amdgpu.ids /tmp/cp$p.err | tail -3 | cut -c1-300
  done
done
