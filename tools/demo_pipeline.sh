#!/bin/bash
# End-to-end run of the reference's workflow on the build's own simulator:
#   collect expert data (scripted pilot) -> behaviour cloning -> HIRL-soft with that bc_actor / TD3 without -> validation.
# Usage: tools/demo_pipeline.sh <scenario> <out_dir> [bc_episodes] [rl_episodes] [num_envs] [seed] [dtype] [replay_rows] [skip_td3]
set -e
set -o pipefail
ENV=${1:-straight_line}; OUT=${2:-gpurun_out/demo}; BC_EP=${3:-200}; RL_EP=${4:-100}; N=${5:-4096}; SEED=${6:-0}; DT=${7:-f32}; BUF=${8:-1048576}; SKIP_TD3=${9:-}
mkdir -p "$OUT"
if [ ! -f "$OUT/../expert_$ENV.csv" ]; then  # one expert set per scenario, shared by the seeds
  python -m hirl4ucav_amd.data.ai_data_col --env "$ENV" --random --episodes 20 --out "$OUT/../expert_$ENV.csv" 2>&1 | tee "$OUT/../collect_$ENV.log"
fi
CSV="$OUT/../expert_$ENV.csv"
python -m hirl4ucav_amd.train_all --agent BC --env "$ENV" --random --seed "$SEED" --episodes "$BC_EP" --checkpoint_rate 50 --bc_validate_from 50 \
    --expert_csv "$CSV" --result_dir "$OUT/results" 2>&1 | grep -v "^Episode .*[1-9]:\|amdgpu.ids" | tee "$OUT/bc.log"
BC_ACTOR=$(ls -t "$OUT"/results/"$ENV"/BC/model/*/model/*Actor_Harfang_GYM | head -1)
echo "bc_actor: $BC_ACTOR" | tee -a "$OUT/bc.log"
python -m hirl4ucav_amd.train_all --agent HIRL --type soft --env "$ENV" --random --seed "$SEED" --episodes "$RL_EP" --num_envs "$N" --snapshot_every 0 --dtype "$DT" --buffer_size "$BUF" \
    --expert_csv "$CSV" --bc_actor "$BC_ACTOR" --result_dir "$OUT/results" 2>&1 | grep -v "amdgpu.ids" | tee "$OUT/hirl_soft.log"
[ -n "$SKIP_TD3" ] || python -m hirl4ucav_amd.train_all --agent TD3 --env "$ENV" --random --seed "$SEED" --episodes "$RL_EP" --num_envs "$N" --snapshot_every 0 --dtype "$DT" \
    --buffer_size "$BUF" --result_dir "$OUT/results" 2>&1 | grep -v "amdgpu.ids" | tee "$OUT/td3.log"
rm -rf "$OUT/results"  # (checkpoints and event files: not evidence, and gpurun_out/ is capped at 64 MiB)
