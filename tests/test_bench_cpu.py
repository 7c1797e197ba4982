"""bench.py's launcher logic without a GPU: it must refuse to print a smaller run under a bigger --gpus."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_fails_loudly_without_n_gpus():
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "5", "--warmup", "1"], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK")})
    assert p.returncode != 0 and "8 GPUs requested" in p.stderr and p.stdout.strip() == ""


def test_world_size_must_match_gpus():
    env = {**os.environ, "WORLD_SIZE": "2", "RANK": "0"}
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--steps", "5", "--warmup", "1"], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300, env=env)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr


def test_cpu_baseline_legs_run_without_a_gpu():
    sys.path.insert(0, ROOT)
    import bench

    b0 = bench.baseline_reference_plumbing(1.0)
    assert b0["value"] > 0 and "messages per step" in b0["sample"]
    port = bench.baseline_port(bench.parse(["--envs", "512"]), 1.0)
    assert port["kind"] == "port" and port["value"] > 0
    b1 = bench.baseline_batched_cpu(0.5)
    assert b1["value"] > port["value"]
    sac = bench.baseline_port_sac(bench.parse(["--agent", "sac", "--envs", "512", "--scenario", "serpentine"]), 1.0)
    assert sac["kind"] == "port" and sac["value"] > 0 and "SAC" in sac["sample"]


def test_workload_label_names_a_baseline_config_only_when_the_arguments_match():
    sys.path.insert(0, ROOT)
    import bench

    assert "configs[1]" in bench.workload_label(bench.parse([]))
    assert "configs[" not in bench.workload_label(bench.parse(["--envs", "2048"]))
    assert "configs[" not in bench.workload_label(bench.parse(["--actions", "uniform"]))
    assert "configs[3]" in bench.workload_label(bench.parse(["--envs", "8192", "--scenario", "circular", "--type", "linear"]))
    assert "configs[4]" in bench.workload_label(bench.parse(["--envs", "16384", "--scenario", "mixed", "--dtype", "bf16"]))
    assert "configs[4]" not in bench.workload_label(bench.parse(["--envs", "16384", "--scenario", "mixed", "--dtype", "bf16_policy"]))
    assert "configs[2]" in bench.workload_label(bench.parse(["--agent", "sac", "--envs", "16384", "--scenario", "serpentine"]))
