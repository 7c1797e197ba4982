#!/bin/bash
# Does the collapse of the last validation (profiles/r03_demo: seed 1 of every scenario) come from the replay window?  With 4,096 envs stepping
# in phase, a 1M-row ring holds the last 256 vector steps — a sixth of ONE episode; the reference's 1e5 rows hold ~100 episodes of its one env.
# HIRL-soft, seed 1, 200 episodes, ring of 1M rows against 16M rows (2 GB of the 288 GB).   tools/demo_replay_size.sh <tag> [episodes] [seed]
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
cd "$R"
TAG=${1:-r03_replay}; EP=${2:-200}; SEED=${3:-1}
for ENV in straight_line serpentine circular; do
  for BUF in 1048576 16777216; do
    bash tools/demo_pipeline.sh "$ENV" "gpurun_out/$TAG/$ENV/rows$BUF" 200 "$EP" 4096 "$SEED" f32 "$BUF" skip > "gpurun_out/$TAG.$ENV.$BUF.out" 2>&1 || echo "FAILED $ENV $BUF"
    rm -f "gpurun_out/$TAG.$ENV.$BUF.out"
    grep -h "validation\|Validation" "gpurun_out/$TAG/$ENV/rows$BUF/hirl_soft.log" | tail -8 | sed "s/^/$ENV rows=$BUF: /"
  done
done
