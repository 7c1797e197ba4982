#!/bin/bash
# The two-rank launch form of bench.py N times in a row on one GPU box (gloo, both ranks on the one GPU): every run must finish and keep its
# replicas identical.  Round 2 saw this form hang once in a dozen runs (bench.py's settle phase left its loop on a rank-local clock).
#   tools/soak_two_ranks.sh [runs]          -> gpurun_out/soak_two_ranks.log
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
cd "$R"
mkdir -p gpurun_out
HX_SOAK=${1:-30} python3 -m pytest tests/test_bench_gpu.py -m gpu -q -x -k soak 2>&1 | tee gpurun_out/soak_two_ranks.log | tail -5
