#!/bin/bash
# Does the FRONT loop (train_all --loop front: the minibatch drawn from the ring as it stood before the step's insert) learn like the
# reference's order (--loop reference: store, then sample, hirl/train_all.py:343-361)?  A/B on the same binaries: per scenario ONE expert set,
# per seed ONE BC checkpoint, then HIRL-soft and TD3 with both loops.   tools/demo_front_vs_reference.sh <tag> [rl_episodes] [seeds] [scenarios]
#   -> gpurun_out/<tag>/<scenario>/seed<k>/{bc,hirl_soft.front,hirl_soft.reference,td3.front,td3.reference}.log + summary.md
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
cd "$R"
TAG=${1:-r05_demo_front_vs_reference}; EP=${2:-100}; SEEDS=${3:-"0 1 2"}; ENVS=${4:-"straight_line serpentine circular"}
N=4096; BUF=1048576
for ENV in $ENVS; do
  D="gpurun_out/$TAG/$ENV"; mkdir -p "$D"
  CSV="$D/expert_$ENV.csv"
  [ -f "$CSV" ] || python -m hirl4ucav_amd.data.ai_data_col --env "$ENV" --random --episodes 20 --out "$CSV" > "$D/collect_$ENV.log" 2>&1
  for SEED in $SEEDS; do
    OUT="$D/seed$SEED"; mkdir -p "$OUT"
    python -m hirl4ucav_amd.train_all --agent BC --env "$ENV" --random --seed "$SEED" --episodes 200 --checkpoint_rate 50 --bc_validate_from 50 \
        --expert_csv "$CSV" --result_dir "$OUT/results" 2>&1 | grep -v "^Episode .*[1-9]:\|amdgpu.ids" > "$OUT/bc.log"
    BC_ACTOR=$(ls -t "$OUT"/results/"$ENV"/BC/model/*/model/*Actor_Harfang_GYM | head -1)
    echo "bc_actor: $BC_ACTOR" >> "$OUT/bc.log"
    for LOOP in front reference; do
      python -m hirl4ucav_amd.train_all --agent HIRL --type soft --env "$ENV" --random --seed "$SEED" --episodes "$EP" --num_envs "$N" --snapshot_every 0 \
          --buffer_size "$BUF" --loop "$LOOP" --expert_csv "$CSV" --bc_actor "$BC_ACTOR" --result_dir "$OUT/results" 2>&1 | grep -v "amdgpu.ids" > "$OUT/hirl_soft.$LOOP.log" \
          || echo "FAILED $ENV seed $SEED HIRL $LOOP"
      python -m hirl4ucav_amd.train_all --agent TD3 --env "$ENV" --random --seed "$SEED" --episodes "$EP" --num_envs "$N" --snapshot_every 0 \
          --buffer_size "$BUF" --loop "$LOOP" --result_dir "$OUT/results" 2>&1 | grep -v "amdgpu.ids" > "$OUT/td3.$LOOP.log" \
          || echo "FAILED $ENV seed $SEED TD3 $LOOP"
    done
    rm -rf "$OUT/results"  # (checkpoints and event files: not evidence, and gpurun_out/ is capped at 64 MiB)
  done
  rm -f "$CSV"
done
python3 tools/demo_front_summary.py "gpurun_out/$TAG" | tee "gpurun_out/$TAG/summary.md"
