#!/bin/bash
# One round's measurement set in ONE gpurun call -> gpurun_out/<tag>/ (copy what is to be judged into profiles/).
#   tools/measure_round.sh <tag> [quick]
# default line (sweep + CPU baselines; the front loop where it applies, the reference order timed in the same run), driver form, the other BASELINE configs
# (+ --no-front twins), kernel stats under rocprofv3 for fp32 (front and reference order) / bf16 / the sharded
# rank's launch sequence over RCCL at world size 1 (program directly after `--`: python3 <script>).
set -uo pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
TAG=${1:-r06}; QUICK=${2:-}
O="$R/gpurun_out/$TAG"
mkdir -p "$O"
cd "$R"
python3 bench.py > "$O/bench_default_line_with_sweep_and_cpu.json" 2> "$O/bench_default.err"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$O/bench_driver_form_steps20_warmup5.json" 2>/dev/null
: > "$O/bench_other_configs.jsonl"
while read -r cfg; do
  [ -z "$cfg" ] && continue
  # shellcheck disable=SC2086
  python3 bench.py $cfg --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 >> "$O/bench_other_configs.jsonl"
done <<'CFGS'
--no-front
--dtype bf16
--no-front --dtype bf16
--dtype bf16_policy
--dtype f32x9
--no-front --dtype f32x9
--agent sac --envs 16384 --scenario serpentine
--agent sac --envs 16384 --scenario serpentine --no-front
--envs 16384 --scenario mixed
--envs 16384 --scenario mixed --dtype bf16
--envs 16384 --scenario mixed --dtype bf16 --no-front
--envs 8192 --scenario circular --type linear --bc_weight 0.5
--envs 65536 --scenario circular --type linear --bc_weight 0.5
--envs 131072 --scenario mixed --dtype bf16
--envs 131072 --scenario mixed
--envs 8192 --scenario circular --type linear --bc_weight 0.5 --no-front
--actions uniform
--staged
--staged --no-front
--batch 256
--batch 256 --no-front
--batch 512
--batch 1024
--batch 1024 --dtype bf16
CFGS
STEPS=4000
stats() {  # stats <name> <bench flags...>: per-kernel stats of a $STEPS-step run under rocprofv3
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_$name" -- python3 "$R/bench.py" --no-cpu-baseline --no-sweep --steps $STEPS --warmup 500 --settle-s 0.5 "$@" > "$O/bench_${name}_line_under_rocprof.json" 2> "$O/bench_${name}_rocprof.err" )
  cp "$O"/prof_"$name"/*/*kernel_stats.csv "$O/bench_${name}_kernel_stats.csv" 2>/dev/null
  rm -rf "$O/prof_$name"
}
stats f32
stats f32_reference_order --no-front
stats bf16 --dtype bf16
stats staged_f32 --staged
if [ -z "$QUICK" ]; then
  # the configs whose dominant kernel is the persistent acting launch: SAC 16,384 serpentine (configs[2]), 65,536 circular
  # HIRL-linear (configs[3] as one population), 131,072 mixed bf16 (configs[4] as one population), 16,384 mixed bf16 (one GPU's shard of configs[4])
  STEPS=2000
  stats sac16k_serpentine --agent sac --envs 16384 --scenario serpentine
  stats circ65536_f32 --envs 65536 --scenario circular --type linear --bc_weight 0.5
  stats mixed131072_bf16 --envs 131072 --scenario mixed --dtype bf16
  stats mixed16k_bf16 --envs 16384 --scenario mixed --dtype bf16
  stats circ8192_f32 --envs 8192 --scenario circular --type linear --bc_weight 0.5
  stats mixed16k_f32 --envs 16384 --scenario mixed
  STEPS=4000
fi
# the sharded rank's launch sequence with its two messages per actor call through RCCL (world size 1): the launcher is a python -m module, so the
# profiler wraps the RANK (bench.py re-enters itself as the single rank when WORLD_SIZE is set)
( cd /tmp && export TMPDIR=/tmp && WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof_nccl1" -- python3 "$R/bench.py" --gpus 1 --staged --no-cpu-baseline --no-sweep --steps 4000 --warmup 500 --settle-s 0.5 > "$O/bench_staged_nccl_world1_line_under_rocprof.json" 2> "$O/bench_staged_nccl_world1.err" )
cp "$O"/prof_nccl1/*/*kernel_stats.csv "$O/bench_staged_nccl_world1_kernel_stats.csv" 2>/dev/null
rm -rf "$O/prof_nccl1"
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29518 python3 bench.py --gpus 1 --staged --no-cpu-baseline --no-sweep > "$O/bench_staged_nccl_world1.json" 2>/dev/null
# the same through torch.distributed.all_reduce (the round-3 path): what taking the host out of the exchange is worth at world size 1
WORLD_SIZE=1 RANK=0 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29519 python3 bench.py --gpus 1 --staged --exchange rccl-torch --no-cpu-baseline --no-sweep > "$O/bench_staged_nccl_world1_torch_allreduce.json" 2>/dev/null
ls -la "$O"
tail -c 400 "$O/bench_driver_form_steps20_warmup5.json"
