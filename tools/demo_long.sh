#!/bin/bash
# HIRL-soft closer to the reference's regime: few envs, many episodes (the reference: 1 env, 6,000 episodes, one update per transition).
# 64 envs x 2,000 episodes = one update per 64 transitions, 3 M updates, validation every 25 episodes.   tools/demo_long.sh <tag> [episodes] [envs] [seed]
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
cd "$R"
TAG=${1:-r03_long}; EP=${2:-2000}; N=${3:-64}; SEED=${4:-1}
for ENV in straight_line serpentine circular; do
  bash tools/demo_pipeline.sh "$ENV" "gpurun_out/$TAG/$ENV/run" 200 "$EP" "$N" "$SEED" f32 1048576 skip > "gpurun_out/$TAG.$ENV.out" 2>&1 || echo "FAILED $ENV"
  rm -f "gpurun_out/$TAG.$ENV.out"
  grep -h "^Validation" "gpurun_out/$TAG/$ENV/run/hirl_soft.log" | awk '{print $2, $10}' | tr '\n' ' ' | sed "s/^/$ENV (validation: success) /"; echo
  grep -v "^Episode" "gpurun_out/$TAG/$ENV/run/hirl_soft.log" > "gpurun_out/$TAG/$ENV/run/hirl_soft_validations.log"; tail -3 "gpurun_out/$TAG/$ENV/run/hirl_soft.log" >> "gpurun_out/$TAG/$ENV/run/hirl_soft_validations.log"
  rm -f "gpurun_out/$TAG/$ENV/run/hirl_soft.log" "gpurun_out/$TAG/$ENV/expert_$ENV.csv"
done
