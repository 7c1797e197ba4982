#!/bin/bash
# Round-3 verdict item 6, priced: (1) the dispatch side of riding launch A inside the act + env launch (merge_probe.hip: spin kernels with the
# step's workgroup counts and in-workgroup times), (2) the real act + env launch at 4,096 envs with 16-row (256 workgroups) and 32-row (128) workgroups —
# the 32-row form is the only one that leaves CUs free for launch A's workgroups.
cd "$(dirname "$0")/../.."
tools/ubench/merge_probe.bin
for rows in 8192 4096; do
  echo "HX_ACT_F32_NRT2_ROWS=$rows (4,096 envs: $([ $rows = 4096 ] && echo 32-row, 128 workgroups || echo 16-row, 256 workgroups))"
  HX_ACT_F32_NRT2_ROWS=$rows SIZES=4096 python tools/ubench/actp_time.py f32 2>&1 | tail -1
done
