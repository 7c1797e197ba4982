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


def test_loop_choice_never_takes_in_launch_waits_on_a_shared_gpu():
    """ranks that share a device run the reference's order unless --front asks for the other; every refusal names its reason."""
    sys.path.insert(0, ROOT)
    import bench

    front, why = bench.loop_choice(bench.parse([]), 1, False)
    assert front and "default" in why
    front, why = bench.loop_choice(bench.parse(["--gpus", "8"]), 8, True)
    assert not front and "8 ranks share a GPU" in why
    front, why = bench.loop_choice(bench.parse(["--gpus", "8"]), 8, False)  # a GPU each: the front loop, sharded
    assert front
    front, why = bench.loop_choice(bench.parse(["--gpus", "2", "--front"]), 2, True)  # asked for explicitly (soak tests)
    assert front and "--front given" in why and "ranks share a GPU" in why
    assert bench.loop_choice(bench.parse(["--no-front"]), 1, False) == (False, "--no-front given")
    for argv, word in ((["--batch", "512"], "batch > 256"), (["--actions", "uniform"], "uniform actions"), (["--dtype", "bf16_policy"], "bf16_policy"),
                       (["--agent", "sac"], "beyond 8,192 envs"), (["--overlap"], "--overlap"), (["--sample-launch"], "--sample-launch"),
                       (["--envs", "16384", "--front-acting", "mfma"], "at most 8,192 envs")):
        front, why = bench.loop_choice(bench.parse(argv), 1, False)
        assert not front and word in why, (argv, why)
    assert bench.loop_choice(bench.parse(["--agent", "sac", "--envs", "16384"]), 1, False)[0]
    assert not bench.loop_choice(bench.parse(["--agent", "sac", "--envs", "16384", "--gpus", "2"]), 2, False)[0]
    import pytest

    with pytest.raises(SystemExit, match="batch > 256"):
        bench.loop_choice(bench.parse(["--front", "--batch", "512"]), 1, False)


def test_roofline_fractions_stay_inside_the_roof_they_name():
    """the exact-split launches are priced by the bf16 FLOPs they execute; round 5's 131,072-env fp32 line (117.9 us at 65,536 envs,
    242 us at 131,072) read `frac` 0.93 / 1.04 against the fp32 matrix peak."""
    sys.path.insert(0, ROOT)
    from tools import bench_roofline as RL

    for envs, us in ((4096, 21.6), (65536, 117.86), (131072, 228.0)):
        flop = RL.ACTOR_FLOP * envs
        r = RL.matrix_roof(flop, envs, us, "x9")
        assert r["peak"] == 2500.0 and 0 < r["frac"] < 1 and r["flop_per_launch"] == flop + 5 * RL.PRODUCT_FLOP * envs
        eq = r["fp32_equivalent"]
        assert eq["peak"] == 157.3 and abs(eq["ratio_to_peak"] - flop / us / 1e6 / 157.3) < 1e-3 and "frac" not in eq
    assert RL.matrix_roof(RL.ACTOR_FLOP * 131072, 131072, 228.0, "x9")["fp32_equivalent"]["ratio_to_peak"] > 0.99  # what used to be printed as frac
    assert RL.matrix_roof(RL.ACTOR_FLOP * 4096, 4096, 19.2, "f32")["peak"] == 157.3 and RL.matrix_roof(1e9, 4096, 19.2, "bf16")["peak"] == 2500.0
