"""bench.py end to end on a GPU box: the single-GPU line and the N > 1 launch form the driver uses (two ranks sharing one GPU over
gloo — HX_BENCH_BACKEND exists for exactly this), checked against the JSON contract."""
import json
import os
import signal
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
            "data", "config", "roofline")


def run(cmd, env=None):
    # One retry on a TIMEOUT only (never on a wrong result): two processes rendezvousing over loopback and sharing one GPU hung once in a dozen
    # suite runs on one box (600 s without output, the same command 5 s four times in a row on the next box); a deadlock of ours would repeat.
    for attempt in (0, 1):
        proc = subprocess.Popen(cmd, cwd=ROOT, env={**os.environ, **(env or {})}, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                start_new_session=True)
        try:
            out, err = proc.communicate(timeout=300)
            break
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)  # the launcher AND its ranks
            proc.communicate()
            if attempt == 1:
                raise
            print("bench.py timed out after 300 s: one retry", file=sys.stderr)
    p = subprocess.CompletedProcess(cmd, proc.returncode, out, err)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]  # ONE JSON line, from rank 0
    return json.loads(lines[0])


def check(d, n_gpus, steps, warmup):
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - 4096 * n_gpus * 1e3 / d["ms_per_step"]) < 1e-3 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["launches_timed"] > 0 and (r["traffic"] is None or r["traffic"] > 0)
    assert "settle_s" in d and "timed_region" in d and d["stage_us"]["sample+learn"] > 0


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_bench_single_gpu_line():
    d = run([sys.executable, "bench.py", "--steps", "120", "--warmup", "20", "--cpu-seconds", "6", "--settle-s", "0.3"])
    check(d, 1, 120, 20)
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    # reference plumbing (one socket env), batched CPU integrator, eager ROCm learn()
    assert 0 < c["b0_reference_plumbing"]["value"] < 5000 and c["b1_batched_cpu"]["value"] > c["value"] and c["b2_eager_rocm_learn"]["value"] > 0
    sw = d["roofline_env_sweep"]
    assert [r["envs_per_launch"] for r in sw] == [4096, 65536, 1 << 20, 1 << 22] and sw[2]["frac"] > 0.4  # the >= 40 % HBM evidence
    assert d["roofline"]["traffic"] is None and "traffic_from_profiles" in d["roofline"]


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus 2` on a 1-GPU box must fail loudly, not print an N = 1 number."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has several GPUs")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "1"], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and "2 GPUs requested" in p.stderr and "{" not in p.stdout


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher environment starts the two ranks itself (here over gloo on the shared GPU)."""
    d = run([sys.executable, "bench.py", "--gpus", "2", "--steps", "40", "--warmup", "5", "--no-cpu-baseline", "--no-sweep", "--settle-s", "0.2"],
            env={"HX_BENCH_BACKEND": "gloo"})
    check(d, 2, 40, 5)
    assert d["rccl_ranks"]["world_size"] == 2 and d["replicas_identical"] is True


# one-shot exchange with both ranks on ONE GPU: the waiting rank's kernel shares the chip with the peer's launches — with 16 workgroups
# (OneShotExchange picks that when ranks share a device; 256 spinning workgroups keep the peer's 1024-thread workgroups from being placed
# for seconds at a time) — a functional check with a long timeout; ranks with a GPU each never wait like that
ONESHOT = ["--exchange", "oneshot", "--exchange-timeout-ms", "120000", "--measure-steps", "16"]


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.parametrize("extra", [[], ["--overlap"], ["--agent", "sac", "--scenario", "serpentine"], ONESHOT])
def test_bench_two_ranks_launch_form(extra):
    """python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ..."""
    port = str(29600 + (os.getpid() + len(extra)) % 300)
    steps, warm = (("12", "2") if "oneshot" in extra else ("60", "10"))
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
             "bench.py", "--gpus", "2", "--steps", steps, "--warmup", warm, "--no-cpu-baseline", "--no-sweep", "--settle-s", "0" if "oneshot" in extra else "0.2"] + extra,
            env={"HX_BENCH_BACKEND": "gloo"})
    check(d, 2, int(steps), int(warm))
    assert d["allreduce"] and all(a["median_us"] > 0 for a in d["allreduce"])
    assert d["env_stats"]["env_steps"] > 0
    assert d["replicas_identical"] is True  # 70 sharded updates later every rank holds the same networks and Adam moments, bit for bit
    assert d["rccl_ranks"]["exchange"] == ("oneshot" if "oneshot" in extra else "rccl") and d["rccl_ranks"]["world_size"] == 2
    if "sac" not in extra:  # HIRL: one message per phase = the critic's 1.1 MB and the merged actor message, nothing else
        assert len(d["allreduce"]) == 2, d["allreduce"]


@pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
@pytest.mark.parametrize("agent", [["--agent", "HIRL", "--type", "soft", "--env", "straight_line"], ["--agent", "SAC", "--type", "SAC", "--env", "serpentine"]])
def test_driver_two_ranks(agent, tmp_path):
    """python -m torch.distributed.run --nproc-per-node 2 -m hirl4ucav_amd.train_all ...: env shards, all-reduced gradients, the replica
    check at the validation episode, checkpoints from rank 0 only."""
    port = str(29300 + os.getpid() % 300)
    code = ("import sys; from hirl4ucav_amd import train_all as T; T.MAX_STEP['straight_line'] = T.MAX_STEP['serpentine'] = 40; "
            "T.main(T.parser().parse_args(sys.argv[1:]))")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", port, "--no-python", sys.executable, "-c", code] + agent +
                       ["--random", "--seed", "1", "--num_envs", "512", "--episodes", "2", "--checkpoint_rate", "2", "--snapshot_every", "0", "--synthetic_expert",
                        "--buffer_size", "65536", "--result_dir", str(tmp_path)],
                       cwd=ROOT, env={**os.environ, "HX_DIST_BACKEND": "gloo"}, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "Episode 2:" in p.stdout and "Validation 1:" in p.stdout and "diverged" not in p.stderr
    assert p.stdout.count("Episode 2:") == 1  # rank 0 reports
