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
            "data", "config", "roofline", "front_status", "update_steps_per_s", "update_samples_per_s")


def rank_lines(err):
    """the ONE line each failing rank writes (`bench.py rank R: Type: message`): torchrun's failure table fills the tail of stderr completely, and the
    rank's own traceback is what a failure report needs first (GPUTEST_r05 showed only the table)"""
    return "\n".join(ln for ln in err.splitlines() if ln.startswith("bench.py rank "))


def launch(cmd, env=None, timeout=300):
    # No retry: the one hang this suite ever saw (two ranks, 600 s without output, once in a dozen runs) was bench.py's own settle phase —
    # each rank left its loop on a LOCAL clock test, so one rank could start the warm-up barrier while its peer enqueued one more flag
    # all-reduce: mismatched collective sequences.  The loop now leaves only on the all-reduced flag (bench.py settle); a timeout here
    # is a failure, reported with both ranks' output.
    proc = subprocess.Popen(cmd, cwd=ROOT, env={**os.environ, **(env or {})}, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                            start_new_session=True)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(proc.pid, signal.SIGKILL)  # the launcher AND its ranks
        out, err = proc.communicate()
        raise AssertionError(f"did not finish within {timeout} s\n{rank_lines(err)}\nstdout: {out[-2000:]}\nstderr: {err[-4000:]}")
    assert proc.returncode == 0, f"exit {proc.returncode}\n{rank_lines(err)}\n--- stderr tail ---\n{err[-3000:]}"
    return out, err


def run(cmd, env=None):
    out, _ = launch(cmd, env)
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]  # ONE JSON line, from rank 0
    assert [l for l in out.splitlines() if l.strip()][-1] == lines[0], out[-600:]  # ... and nothing after it (RCCL's banner comes first)
    return json.loads(lines[0])


def run_many(argvs, nproc=0, env=None, timeout=600):
    """several bench.py argument sets in ONE set of processes (tools/bench_many.py: a cold `import torch` costs ~10 s per process on a fresh box);
    nproc > 0: under torch.distributed.run with that many ranks.  -> the records, in order"""
    head = [sys.executable]
    if nproc:
        head += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1", "--master-port",
                 str(29600 + (os.getpid() + 7 * len(argvs) + nproc) % 300)]
    out, err = launch(head + [os.path.join("tools", "bench_many.py"), json.dumps(argvs)], env, timeout)
    lines = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    assert len(lines) == len(argvs) and all("record" in l for l in lines), (rank_lines(err), [l.get("error") for l in lines], err[-2000:])
    assert [l["argv"] for l in lines] == argvs
    return [l["record"] for l in lines]


def check_roof(r):
    """`frac` is a fraction of a roof the kernel HAS — strictly inside (0, 1) at every size, for every format"""
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == ("GB/s" if r["bound"] == "hbm" else "TFLOP/s")
    assert r["peak"] in (8000.0, 157.3, 2500.0) and 0 < r["frac"] < 1 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    if r.get("acting_format") == "x9" and r["bound"] == "mfma":  # the exact-split product: EXECUTED bf16 FLOPs against the bf16 peak ...
        eq = r["fp32_equivalent"]  # ... the fp32 arithmetic delivered beside it, as a ratio that may pass 1
        assert r["peak"] == 2500.0 and eq["peak"] == 157.3 and r["flop_per_launch"] > 4 * eq["flop_per_launch"] and eq["ratio_to_peak"] > 0
        assert abs(eq["ratio_to_peak"] - eq["achieved"] / 157.3) < 1e-3 and "frac" not in eq


def check(d, n_gpus, steps, warmup, envs=4096, dtype="f32", fused=True, shared_gpu=False, batch=128):
    """shared_gpu: several ranks on ONE GPU (the rehearsals) — launches of different ranks interleave on the chip, so relations between the stamped
    durations of two kernels say nothing there"""
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["steps"] == steps and d["warmup"] == warmup and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["dtype"] == dtype and d["data"] == "synthetic" and "workload" in d["config"]
    assert d["value"] > 0 and abs(d["value"] - envs * n_gpus * 1e3 / d["ms_per_step"]) < 1e-3 * d["value"]
    assert abs(d["update_samples_per_s"] - batch * 1e3 / d["ms_per_step"]) < 2e-3 * d["update_samples_per_s"] + 0.1  # (both rounded to 0.1)
    # R = 3 timed repetitions, the line is the median one
    rp = d["repetitions"]
    assert rp["count"] == 3 and len(rp["ms_per_step"]) == 3 and rp["statistic"] == "median"
    assert abs(sorted(rp["ms_per_step"])[1] - d["ms_per_step"]) < 1e-9 and abs(sorted(rp["value"])[1] - d["value"]) < 1e-3 * d["value"]
    r = d["roofline"]
    check_roof(r)
    assert r["launches_timed"] > 0 and (r["traffic"] is None or r["traffic"] > 0) and r["us_per_launch"] > 0
    assert d["config"]["loop"] in ("front", "reference order", "reference order (front tripped)") and d["config"]["loop_reason"]
    assert (d["front_status"] != 0) == (d["config"]["loop"] == "reference order (front tripped)")
    front = d["config"]["loop"] == "front"
    if shared_gpu:  # ranks that share a device never take the in-launch waits by default (GPUTEST_r05)
        assert not front or "--front given" in d["config"]["loop_reason"]
    if front:  # the default loop where it applies: HIRL, one process per GPU — env step + launches A and B of learn() in ONE launch
        assert fused and "_front_kernel" in r["kernel"] and "draw" in d["config"]
        assert shared_gpu or [v for k, v in d["stage_us"].items() if k.startswith("front launch + rest of learn()")][0] >= r["us_per_launch"]
        if n_gpus == 1 and "rccl_ranks" not in d:  # the same workload in the reference's order, timed in the same process
            ro = d["reference_order"]
            assert ro["value"] > 0 and abs(ro["value"] - envs * 1e3 / ro["ms_per_step"]) < 1e-3 * ro["value"] and ro["repetitions"]["count"] == 3
    if fused:  # the record describes the kernel the timed loop RUNS: act + env step in one launch, both roofs quoted
        assert front or (("act_persist_" if envs > 8192 else "act_fused_kernel") in r["kernel"] and "ENV" in r["kernel"])
        # ... and it is the DOMINANT kernel of the step: no other stage's launch outlasts it
        st = d["stage_us"]
        assert shared_gpu or all(v is None or r["us_per_launch"] >= 0.8 * v for k, v in st.items() if k.startswith(("act(", "env_step(")))
        check_roof(r["other_roof"])
        assert {r["bound"], r["other_roof"]["bound"]} == {"hbm", "mfma"} and r["frac"] >= r["other_roof"]["frac"]
        e = d["roofline_env_kernel"]
        assert e["bound"] == "hbm" and "env_step_kernel" in e["kernel"] and 0 < e["frac"] < 1 and "traffic_from_profiles" in e
        assert shared_gpu or r["us_per_launch"] > e["us_per_launch"] * 0.8  # the fused launch contains the env step
        assert front or shared_gpu or d["stage_us"]["act+env_step(1 kernel)"] >= r["us_per_launch"] * 0.8  # events around the launch >= the kernel's own stamps
    else:
        assert r["bound"] == "hbm" and "env_step_kernel" in r["kernel"] and "roofline_env_kernel" not in d
    if "roofline_act" in d:
        check_roof(d["roofline_act"])
    if "roofline_update" in d:
        check_roof(d["roofline_update"])
    assert "settle_s" in d and "dry_regions" in d and "timed_region" in d and d["stage_us"]["sample+learn"] > 0


needs_gpu = pytest.mark.skipif(not torch.cuda.is_available(), reason="needs a GPU")
FAST = ["--steps", "60", "--warmup", "10", "--no-sweep", "--no-cpu-baseline", "--settle-s", "0.2"]


@needs_gpu
def test_bench_single_gpu_line():
    d = run([sys.executable, "bench.py", "--steps", "120", "--warmup", "20", "--cpu-seconds", "6", "--settle-s", "0.3"])
    check(d, 1, 120, 20)
    assert "configs[1]" in d["config"]["workload"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and c["sample"]
    # reference plumbing (one socket env), batched CPU integrator, eager ROCm learn()
    assert 0 < c["b0_reference_plumbing"]["value"] < 5000 and c["b1_batched_cpu"]["value"] > c["value"] and c["b2_eager_rocm_learn"]["value"] > 0
    sw = d["roofline_env_sweep"]
    assert [r["envs_per_launch"] for r in sw] == [4096, 65536, 1 << 20, 1 << 22] and sw[2]["frac"] > 0.4  # the >= 40 % HBM evidence
    # the default loop on a GPU of its own: the front launch, in the exact-split acting format, priced against the bf16 peak it executes on
    assert d["config"]["loop"] == "front" and d["front_status"] == 0 and d["roofline"]["acting_format"] == "x9" and d["roofline"]["peak"] == 2500.0
    # traffic: the committed PMC passes of THIS kernel at THIS size (fp32 HIRL, 4,096 envs), labelled as a profile artefact; never invented
    tp = d["roofline"]["traffic_from_profiles"]
    assert d["roofline"]["traffic"] == tp["bytes"] == tp["fetch_bytes"] + tp["write_bytes"] and tp["source"] == "profiles/pmc_env_traffic.json"
    assert d["roofline_env_kernel"]["traffic"] == d["roofline_env_kernel"]["traffic_from_profiles"]["bytes"]


@needs_gpu
def test_bench_labels_follow_the_arguments():
    """The workload string is built from the run's own arguments; a BASELINE.json configs[] index appears only where they match it; no
    `roofline_act` is computed from a random-number fill; SAC lines carry a CPU baseline too; the large-population and large-batch lines keep
    every `frac` inside (0, 1).  Nine argument sets in one process (tools/bench_many.py)."""
    nocpu = [a for a in FAST]
    sets = [["--envs", "8192", "--scenario", "circular", "--type", "linear", "--bc_weight", "0.5"] + nocpu,
            ["--no-front"] + nocpu,  # the reference's order on every step: act + env step as a launch of their own
            ["--actions", "uniform"] + nocpu,
            ["--envs", "16384", "--scenario", "mixed", "--dtype", "bf16"] + nocpu,
            ["--dtype", "f32x9"] + nocpu,
            ["--agent", "sac", "--envs", "16384", "--scenario", "serpentine", "--cpu-seconds", "4"] + [a for a in FAST if a != "--no-cpu-baseline"],
            ["--envs", "65536", "--scenario", "circular", "--type", "linear"] + nocpu,  # round 5's convention read frac 0.93 here, 1.04 at 131,072
            ["--envs", "131072", "--scenario", "mixed"] + nocpu,
            ["--batch", "512"] + nocpu]  # M2 at a large batch: beyond 256 rows the reference's order
    d8, dn, du, db, dx, ds, d64, d128, dB = run_many(sets)
    check(d8, 1, 60, 10, envs=8192)
    w = d8["config"]["workload"]
    assert "configs[3]" in w and "configs[1]" not in w and "HIRL-linear" in w and "8192 parallel circular" in w
    check(dn, 1, 60, 10)
    assert dn["config"]["loop"] == "reference order" and dn["config"]["loop_reason"] == "--no-front given"
    assert "act_fused_kernel" in dn["roofline"]["kernel"] and "reference_order" not in dn
    check(du, 1, 60, 10, fused=False)
    assert "roofline_act" not in du and "configs[" not in du["config"]["workload"] and "uniform actions" in du["config"]["loop_reason"]
    check(db, 1, 60, 10, envs=16384, dtype="bf16")  # > 8,192 envs: the persistent acting kernel, still ONE launch with the env step
    assert "configs[4]" in db["config"]["workload"] and "bf16 actor/critic" in db["config"]["workload"]
    assert db["roofline_update"]["peak"] == 2500.0 and db["roofline_act"]["peak"] == 2500.0 and db["roofline"]["acting_format"] == "bf16"
    check(dx, 1, 60, 10, dtype="f32 (policy product: exact three-way bf16 split, six partial products)")  # opt-in: never labelled as the configs[1] line
    assert "configs[" not in dx["config"]["workload"] and "three-way bf16 split" in dx["config"]["workload"]
    assert dx["roofline"]["peak"] == 2500.0 and dx["roofline"]["traffic"] is None  # (the committed PMC passes are of the default-format launch)
    check(ds, 1, 60, 10, envs=16384)
    assert "configs[2]" in ds["config"]["workload"] and ds["cpu_baseline"]["kind"] == "port" and ds["cpu_baseline"]["value"] > 0
    assert "SAC" in ds["cpu_baseline"]["sample"] and ds["cpu_baseline"]["b1_batched_cpu"]["value"] > 0 and ds["roofline"]["acting_format"] == "x9"
    check(d64, 1, 60, 10, envs=65536)
    check(d128, 1, 60, 10, envs=131072)
    for d in (d64, d128):  # the persistent exact-split kernels: executed bf16 FLOPs; the fp32-equivalent ratio is what used to be printed as `frac`
        assert d["roofline"]["acting_format"] == "x9" and d["roofline"]["bound"] == "mfma" and d["roofline"]["fp32_equivalent"]["ratio_to_peak"] > 0.5
    check(dB, 1, 60, 10, batch=512)
    assert dB["config"]["loop"] == "reference order" and "batch > 256" in dB["config"]["loop_reason"] and dB["config"]["batch"] == 512
    assert dB["roofline_update"]["samples_per_s"] > 0 and "configs[" not in dB["config"]["workload"]


@needs_gpu
def test_bench_front_trip_falls_back_in_process():
    """a front launch whose in-launch wait gives up must cost a repeat, not the run.  The trip is injected where a workgroup would
    leave it (the sticky status word, after the timed region); bench.py must repeat the run in the reference's order in the same process, exit 0 and
    say so in the line."""
    d = run([sys.executable, "bench.py", "--inject-front-trip"] + FAST)
    check(d, 1, 60, 10)
    assert d["config"]["loop"] == "reference order (front tripped)" and d["front_status"] == 1 and "tripped" in d["config"]["loop_reason"]
    assert "act_fused_kernel" in d["roofline"]["kernel"] and "reference_order" not in d


@needs_gpu
def test_bench_one_rank_over_rccl_runs_the_sharded_sequence():
    """python -m torch.distributed.run --nproc-per-node 1 ... --gpus 1 --staged with the DEFAULT backend (nccl = RCCL on ROCm): the
    process group is created on the device, the engine runs the sharded rank's launch sequence and sends its two messages per actor
    call through RCCL at world size 1 — by default with ncclAllReduce enqueued by the library itself on the engine's stream (hx_rccl_*:
    the communicator id travels over the process group once), with --exchange rccl-torch through torch.distributed.all_reduce, with --exchange
    rccl-bf16 as bf16 on the wire.  RCCL loads, builds a communicator on an MI355X and accepts the flat gradient messages; what this box cannot
    show is an exchange between two GPUs.  The direct path must not be slower than torch.distributed's (it exists to take ~8 us of host time per
    message out).  Three argument sets on one process group."""
    common = ["--gpus", "1", "--staged"] + FAST
    d, t, h = run_many([common, common + ["--exchange", "rccl-torch"], common + ["--exchange", "rccl-bf16"]], nproc=1, env={"HX_BENCH_BACKEND": "nccl"})
    check(d, 1, 60, 10)
    rr = d["rccl_ranks"]
    assert rr["backend"] == "rccl-direct" and rr["process_group_backend"] == "nccl" and rr["world_size"] == 1 and rr["exchange"] == "rccl-direct" and rr["rccl_version"]
    assert "staged" in d["config"]["update_path"]
    assert len(d["allreduce"]) == 2 and {a["bytes"] for a in d["allreduce"]} == {4 * 276488, 4 * (2 * 138756 + 64)}  # critic message, merged actor message
    assert "world size 1" in d["allreduce_note"]
    assert t["rccl_ranks"]["backend"] == "nccl" and t["rccl_ranks"]["exchange"] == "rccl"
    assert d["ms_per_step"] <= t["ms_per_step"] * 1.03, (d["ms_per_step"], t["ms_per_step"])
    check(h, 1, 60, 10)
    assert h["rccl_ranks"]["exchange"] == "rccl-direct-bf16" and len(h["allreduce"]) == 2


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus 2` on a 1-GPU box must fail loudly, not print an N = 1 number."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has several GPUs")
    p = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "5", "--warmup", "1"], cwd=ROOT, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=300)
    assert p.returncode != 0 and "2 GPUs requested" in p.stderr and "{" not in p.stdout


@needs_gpu
@pytest.mark.skipif(not os.environ.get("HX_SOAK"), reason="soak loop: HX_SOAK=<runs> (tools/soak_two_ranks.sh; 30 runs took ~6 min on one MI355X)")
def test_two_rank_launch_soak():
    """The two-rank launch form N times in a row (the hang of round 2 showed once in a dozen runs): every run must finish."""
    for k in range(int(os.environ["HX_SOAK"])):
        d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
                 str(29400 + (os.getpid() + k) % 500), "bench.py", "--gpus", "2", "--steps", "40", "--warmup", "5", "--reps", "1", "--no-cpu-baseline",
                 "--no-sweep", "--settle-s", "0.3", "--measure-steps", "16"], env={"HX_BENCH_BACKEND": "gloo"})
        assert d["replicas_identical"] is True, k


@needs_gpu
@pytest.mark.skipif(not os.environ.get("HX_SOAK"), reason="opt-in: eight FRONT-loop ranks on one GPU (HX_SOAK=1) — 1,024 waiting workgroups on a 256-CU chip")
def test_front_loop_under_contention_gives_a_valid_line_either_way():
    """The shape that failed on the driver's box in round 5, asked for explicitly (--front): whether the waits hold or trip, the run must end with a
    valid line and exit code 0 — `front`, or `reference order (front tripped)` with the status word."""
    d = run([sys.executable, "bench.py", "--gpus", "8", "--front"] + FAST, env={"HX_BENCH_BACKEND": "gloo"})
    check(d, 8, 60, 10, shared_gpu=True)
    assert d["config"]["loop"] in ("front", "reference order (front tripped)") and d["replicas_identical"] is True


# peer-read exchanges with both ranks on ONE GPU: the waiting rank's kernel shares the chip with the peer's launches — with 16 workgroups
# (OneShotExchange picks that when ranks share a device; 256 spinning workgroups keep the peer's 1024-thread workgroups from being placed
# for seconds at a time) — a functional check with a long timeout; ranks with a GPU each never wait like that
PEER = ["--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--no-sweep", "--settle-s", "0", "--exchange-timeout-ms", "120000", "--measure-steps", "16"]
GLOO = {"HX_BENCH_BACKEND": "gloo"}
# the FRONT loop on a shared GPU, asked for explicitly: short on purpose — a box on which its waits DID run into their ~1 s bound pays that per launch
SHORT = ["--steps", "20", "--warmup", "4", "--no-cpu-baseline", "--no-sweep", "--settle-s", "0", "--measure-steps", "16"]


def check_two_ranks(d, steps, warm, exchange, world=2, sac=False):
    check(d, world, steps, warm, shared_gpu=True)  # the ranks share the box's one GPU: their launches interleave, durations of two kernels do not compare
    assert d["config"]["loop"].startswith("reference order") or "--front given" in d["config"]["loop_reason"]
    assert d["allreduce"] and all(a["median_us"] > 0 for a in d["allreduce"]) and d["env_stats"]["env_steps"] > 0
    assert d["replicas_identical"] is True  # after every sharded update so far each rank holds the same networks and Adam moments, bit for bit
    rr = d["rccl_ranks"]
    assert rr["exchange"] == exchange and rr["world_size"] == world and rr["distinct_gpus"] == 1
    if not sac:  # HIRL: one message per phase = the critic's 1.1 MB and the merged actor message, nothing else
        assert len(d["allreduce"]) == 2, d["allreduce"]


@needs_gpu
def test_bench_two_ranks_launch_form():
    """python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...: the driver's own
    command line at N = 2 (gloo: both ranks on the box's one GPU).  Ranks that share a device take the reference's order, and say why."""
    port = str(29600 + os.getpid() % 300)
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
             "bench.py", "--gpus", "2"] + FAST, env=GLOO)
    check_two_ranks(d, 60, 10, "rccl")
    assert d["config"]["loop"] == "reference order" and "ranks share a GPU" in d["config"]["loop_reason"] and "exchange_probe" not in d


@needs_gpu
def test_bench_two_rank_variants_on_one_process_group():
    """The other two-rank forms as argument sets of ONE launch (tools/bench_many.py): two streams, SAC, the three peer-read exchanges, the FRONT
    loop asked for explicitly (two ranks: 256 waiting workgroups — the soaks of round 5 ran three such processes), an injected trip on the LAST rank
    only (every rank must repeat in the reference's order: one decision for all), and `--exchange auto` — the probe that times BOTH transports
    before the timed loop (rccl in process, twostage in child processes) and takes the faster."""
    sets = [FAST + ["--gpus", "2", "--overlap"], FAST + ["--gpus", "2", "--agent", "sac", "--scenario", "serpentine"],
            PEER + ["--gpus", "2", "--exchange", "oneshot"], PEER + ["--gpus", "2", "--exchange", "twostage"],
            PEER + ["--gpus", "2", "--exchange", "twostage-bf16"], SHORT + ["--gpus", "2", "--front"],
            SHORT + ["--gpus", "2", "--front", "--inject-front-trip"], PEER + ["--gpus", "2", "--exchange", "auto", "--probe-messages", "8"]]
    ov, sac, one, two, twob, fr, trip, auto = run_many(sets, nproc=2, env=GLOO, timeout=900)
    check_two_ranks(ov, 60, 10, "rccl")
    assert ov["config"]["issue_order"] == "two streams"
    check_two_ranks(sac, 60, 10, "rccl", sac=True)
    for d, name in ((one, "oneshot"), (two, "twostage"), (twob, "twostage-bf16")):
        check_two_ranks(d, 12, 2, name)
    check_two_ranks(fr, 20, 4, "rccl")
    assert fr["config"]["loop"] in ("front", "reference order (front tripped)")  # asked for: taken; a trip would have been survived
    check_two_ranks(trip, 20, 4, "rccl")
    assert trip["config"]["loop"] == "reference order (front tripped)" and trip["front_status"] == 1
    pr = auto["exchange_probe"]
    assert pr["requested"] == "auto" and pr["chosen"] in ("rccl", "twostage") and pr["message_bytes"] == 4 * 276488 and pr["messages"] == 8
    names = [t["transport"] for t in pr["transports"]]
    assert names == ["torch.distributed (gloo)", "twostage"] and all(t["ok"] and t["median_us"] > 0 for t in pr["transports"]), pr
    assert pr["transports"][1]["distinct_gpus"] == 1 and len(pr["transports"][1]["per_rank_median_us"]) == 2
    assert pr["fastest"] in names and pr["chosen"] == ("twostage" if pr["fastest"] == "twostage" else "rccl")
    check_two_ranks(auto, 12, 2, pr["chosen"])


@needs_gpu
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
    assert "vector loop: reference order" in p.stdout  # two ranks on one GPU: no in-launch waits


# ---- the 8-rank forms (BASELINE.json configs[3] / configs[4]: 8 x MI355X), rehearsed with all ranks on the box's one GPU ---------------------------------
# No 8-GPU node is available to this suite; what CAN run before the first scaling measurement is everything except the wire: the launcher at 8, eight
# env shards with their own rings, kMaxWorld flag / red arrays and seven hipIpc mappings per rank, the 8-way slices of the two-stage exchange, the
# replica check over eight checksums.  rccl_ranks.distinct_gpus says 1 — that is the point of recording it.  Ranks that share a device run the
# reference's order (bench.loop_choice): eight front-loop ranks are 1,024 waiting workgroups on a 256-CU chip — the shape include/hirl4ucav.h excludes
# and the one that took round 5's suite down on the driver's box.


@needs_gpu
def test_bench_eight_ranks_on_one_gpu():
    """`python bench.py --gpus 8` with no launcher environment: the bench starts its eight ranks itself (gloo for the RCCL-shaped path); the JSON line is
    kept under gpurun_out/ for profiles/."""
    d = run([sys.executable, "bench.py", "--gpus", "8"] + FAST, env=GLOO)
    check_two_ranks(d, 60, 10, "rccl", world=8)
    assert d["config"]["loop"] == "reference order" and "8 ranks share a GPU" in d["config"]["loop_reason"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_bench_8ranks_one_gpu_gloo.json"), "w") as f:
        json.dump(d, f)


@needs_gpu
def test_bench_eight_ranks_two_stage_exchange():
    """the same eight ranks through the peer-read kernels (seven hipIpc mappings per rank, the 8-way slices of hx_allreduce_twostage)"""
    d, = run_many([PEER + ["--gpus", "8", "--exchange", "twostage"]], nproc=8, env=GLOO, timeout=900)
    check_two_ranks(d, 12, 2, "twostage", world=8)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_bench_8ranks_one_gpu_twostage.json"), "w") as f:
        json.dump(d, f)


@needs_gpu
def test_driver_eight_ranks_started_by_the_driver(tmp_path):
    """`python -m hirl4ucav_amd.train_all --gpus 8 ...` without a launcher environment: the driver starts its eight ranks as a child (launch_ranks), each
    with its env shard (env ids [r n, (r + 1) n)) and ring; 10 short episodes of configs[3]'s agent (HIRL-linear, circular) with the replica check every 5."""
    p = subprocess.run([sys.executable, "-m", "hirl4ucav_amd.train_all", "--gpus", "8", "--agent", "HIRL", "--type", "linear", "--env", "circular", "--seed", "1",
                        "--num_envs", "512", "--episodes", "10", "--max_step", "24", "--checkpoint_rate", "10", "--snapshot_every", "5", "--synthetic_expert",
                        "--buffer_size", "65536", "--result_dir", str(tmp_path)],
                       cwd=ROOT, env={**os.environ, "HX_DIST_BACKEND": "gloo"}, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    assert p.stdout.count("Episode 10:") == 1 and "Validation 1:" in p.stdout and "diverged" not in p.stderr
    assert "vector loop: reference order (8 ranks share a GPU" in p.stdout
    run_dir = [dp for dp, _, fs in os.walk(tmp_path) if "log1.txt" in fs]
    assert len(run_dir) == 1  # ONE run directory for all ranks ...
    shards = sorted(f for f in os.listdir(run_dir[0]) if f.startswith("state_rank"))
    assert shards == [f"state_rank{r}.pt" for r in range(8)]  # ... holding every rank's shard of the snapshot
