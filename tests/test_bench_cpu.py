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
