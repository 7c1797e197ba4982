set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m
mkdir -p $O
python3 bench.py > $O/bench_default_line_with_sweep_and_cpu.json 2> $O/bench_default.err
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_driver_form_steps20_warmup5.json 2>/dev/null
for cfg in "--dtype bf16" "--agent sac --envs 16384 --scenario serpentine" "--envs 16384 --scenario mixed" "--envs 16384 --scenario mixed --dtype bf16" "--envs 8192 --scenario circular --type linear --bc_weight 0.5" "--actions uniform" "--separate-launches" "--sample-launch"; do
  python3 bench.py $cfg --no-cpu-baseline --no-sweep 2>/dev/null | tail -1 >> $O/bench_other_configs.jsonl
done
python3 tools/ubench/stamps.py > $O/stamps.txt 2>&1
python3 tools/ubench/spans.py > $O/spans.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sweep > $GRAFT_REPO_ROOT/$O/bench_default_line_under_rocprof.json 2>/dev/null
cd $GRAFT_REPO_ROOT
cp $O/prof/*/*kernel_stats.csv $O/bench_default_kernel_stats.csv
rm -rf $O/prof
bash tools/pmc_learn_passes.sh > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_learn_[1-5] > $O/pmc_sq_counters_per_wave.txt
tools/ubench/handoff_probe.bin > $O/ubench_handoff_probe.txt 2>&1
tools/ubench/shared_table_probe.bin > $O/ubench_shared_table_probe.txt 2>&1
tools/ubench/mfma_rate_probe.bin > $O/ubench_mfma_rate_probe.txt 2>&1
tools/ubench/kernarg_probe.bin > $O/ubench_kernarg_probe.txt 2>&1
timeout 120 tools/ubench/icache_probe.bin > $O/ubench_icache_probe.txt 2>&1
timeout 120 tools/ubench/boundary_probe.bin > $O/ubench_boundary_probe.txt 2>&1
ls -la $O
tail -c 600 $O/bench_default_line_with_sweep_and_cpu.json
