#!/bin/bash
# The "does it learn" check of the round's kernels: BC, HIRL-soft and TD3 on the three scenarios x three seeds -> gpurun_out/<tag>/<scenario>/seed<k>/
# and a summary table (tools/demo_summary.py).   tools/demo_all.sh <tag> [rl_episodes] [dtype]
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
cd "$R"
TAG=${1:-r03_demo}; EP=${2:-100}; DT=${3:-f32}
for ENV in straight_line serpentine circular; do
  for SEED in 0 1 2; do
    bash tools/demo_pipeline.sh "$ENV" "gpurun_out/$TAG/$ENV/seed$SEED" 200 "$EP" 4096 "$SEED" "$DT" > "gpurun_out/$TAG.$ENV.$SEED.out" 2>&1 || echo "FAILED $ENV seed $SEED"
    rm -f "gpurun_out/$TAG.$ENV.$SEED.out"
  done
done
python3 tools/demo_summary.py "gpurun_out/$TAG" | tee "gpurun_out/$TAG/summary.md"
