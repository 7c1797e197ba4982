#!/bin/bash
# End-to-end run of the reference's workflow on the build's own simulator:
#   collect expert data (scripted pilot) -> behaviour cloning -> HIRL-soft with that bc_actor / TD3 without -> validation.
# Usage: tools/demo_pipeline.sh <scenario> <out_dir> [bc_episodes] [rl_episodes] [num_envs]
set -e
set -o pipefail
ENV=${1:-straight_line}; OUT=${2:-gpurun_out/demo}; BC_EP=${3:-200}; RL_EP=${4:-100}; N=${5:-4096}
mkdir -p "$OUT"
python -m hirl4ucav_amd.data.ai_data_col --env "$ENV" --random --episodes 20 --out "$OUT/expert.csv" 2>&1 | tee "$OUT/collect.log"
python -m hirl4ucav_amd.train_all --agent BC --env "$ENV" --random --seed 0 --episodes "$BC_EP" --checkpoint_rate 50 --bc_validate_from 50 \
    --expert_csv "$OUT/expert.csv" --result_dir "$OUT/results" 2>&1 | grep -v "^Episode .*[1-9]:\|amdgpu.ids" | tee "$OUT/bc.log"
BC_ACTOR=$(ls -t "$OUT"/results/"$ENV"/BC/model/*/model/*Actor_Harfang_GYM | head -1)
echo "bc_actor: $BC_ACTOR" | tee -a "$OUT/bc.log"
python -m hirl4ucav_amd.train_all --agent HIRL --type soft --env "$ENV" --random --seed 0 --episodes "$RL_EP" --num_envs "$N" --snapshot_every 0 \
    --expert_csv "$OUT/expert.csv" --bc_actor "$BC_ACTOR" --result_dir "$OUT/results" 2>&1 | grep -v "amdgpu.ids" | tee "$OUT/hirl_soft.log"
python -m hirl4ucav_amd.train_all --agent TD3 --env "$ENV" --random --seed 0 --episodes "$RL_EP" --num_envs "$N" --snapshot_every 0 \
    --result_dir "$OUT/results" 2>&1 | grep -v "amdgpu.ids" | tee "$OUT/td3.log"
