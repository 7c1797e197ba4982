#!/usr/bin/env python3
"""bench.py — the hot path end to end on N GPUs of one node (one process per GPU).

One "step" = one pass of the hot path over one batch of synthetic input, per GPU:
    act      clamp(actor(obs) + N(0, 0.1^2), -1, 1) for all envs           (chooseAction, HIRL.py:192-198)
    env      one simulator tick + obs/reward/done for all envs, fused replay insert   (HarfangEnv.step, train_all.py:343-349)
    sample   device-side minibatch draw                                     (buffer.py:45, HIRL.py:249,265)
    learn    one HIRL-soft Agent.learn at B = 128 (critic every call, actor every 2nd, Polyak every 6th)   (HIRL.py:221-334)
i.e. the reference's inner loop (train_all.py:341-361) with the single socket env replaced by `--envs` resident envs.
The update-to-data ratio is a stated design parameter: 1 learn() of the reference's batch per vector step.

Workload (BASELINE.json configs[1]): 4,096 parallel straight_line envs per GPU, HIRL-soft, fp32, synthetic random-init
episodes (random_reset, Philox), synthetic 20,000-row expert set, seeded-init networks.  Prints ONE JSON line (rank 0).

    python bench.py                       # 1 GPU, defaults finish in about a minute (incl. the bounded CPU baselines)
    python bench.py --gpus N ...          # starts N ranks itself (torch.distributed.run as a child process, before this process touches
                                          # a GPU) and relays rank 0's line; fails if fewer than N GPUs are visible
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Shape of a run: [settle: untimed steps for --settle-s seconds] -> W warm-up steps -> [--dry-regions (1) untimed region of the same shape,
declared] -> R = --reps (3) repetitions of { barrier + synchronize -> EXACTLY K steps with nothing but the hot path on the stream ->
synchronize + barrier (max over ranks) }; `value` / `ms_per_step` are the MEDIAN repetition, all R are listed.
Everything that needs events or stamped launches (stage times, the act + env launch's own duration for the roofline, the stand-alone env
kernel, all-reduce times) runs in a SECOND pass after the clock has been read.

Which loop: the FRONT loop (env step + the first two launches of learn() in one launch, HirlEngine.step_learn) where it
applies AND every rank has a GPU of its own; the reference's order otherwise (`config.loop`, `config.loop_reason`).  The front launch waits inside the
launch; if such a wait ever gives up (sticky status word) the run is repeated in the reference's order IN THIS PROCESS and the line says so
(`config.loop = "reference order (front tripped)"`, `front_status`): a valid line and exit code 0, never a crash and never a number from a tripped loop.

The pieces: tools/bench_inputs.py (synthetic inputs), tools/bench_roofline.py (roofline records and their rule), tools/bench_baselines.py (CPU legs).
"""
import argparse
import copy
import ctypes
import json
import os
import socket
import subprocess
import sys
import time
import traceback

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from tools.bench_baselines import (baseline_batched_cpu, baseline_eager_rocm_learn, baseline_port, baseline_port_sac,  # noqa: E402,F401
                                   baseline_reference_plumbing, cpu_baselines)
from tools.bench_inputs import init_params, synthetic_expert  # noqa: E402
from tools import bench_roofline as RL  # noqa: E402

CRITIC_MESSAGE_FLOATS = 276488  # the flat critic gradient: the larger of the two messages of a sharded update


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20000)  # ~2 s of GPU time
    p.add_argument("--warmup", type=int, default=2000)
    p.add_argument("--settle-s", dest="settle_s", type=float, default=1.5,
                   help="untimed steps of the same loop for this many seconds BEFORE the warm-up (clocks, caches, allocator); 0 = off")
    p.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    p.add_argument("--batch", type=int, default=128, help="rows of the minibatch of one learn(); up to 1,024 (beyond 256: the reference's order)")
    p.add_argument("--scenario", default="straight_line", choices=["straight_line", "serpentine", "circular", "mixed"],
                   help="mixed: scenario id = env id mod 3, sorted so that each third of the shard is one scenario (BASELINE.json configs[4])")
    p.add_argument("--type", default="soft", choices=["soft", "linear", "fixed"], help="HIRL BC-weight schedule (train_all.py:328-339)")
    p.add_argument("--bc_weight", type=float, default=0.5, help="linear / fixed: the weight (configs[3]: linear, 0.5)")
    p.add_argument("--agent", default="hirl", choices=["hirl", "sac"], help="sac: BASELINE.json configs[2] (use --envs 16384 --scenario serpentine)")
    p.add_argument("--dtype", default="f32", choices=["f32", "bf16", "bf16_policy", "f32x9"],
                   help="f32x9: fp32 everywhere, the policy's 256->512 product of the ACTING kernel through the exact three-way bf16 split of both "
                        "operands on the bf16 matrix cores at every size (fp32 results up to summation order; f32 takes it from 4,096 rows on); "
                        "bf16: actor AND critic — policy inference and the three 256<->512 products of every network in learn() on bf16 MFMA, fp32 "
                        "accumulation, fp32 master weights / Adam / LayerNorm / dynamics (BASELINE.json configs[4]); bf16_policy: policy inference only")
    p.add_argument("--reps", type=int, default=3, help="timed repetitions of K steps; value = the median repetition (SURVEY.md 8d)")
    p.add_argument("--dry-regions", dest="dry_regions", type=int, default=1,
                   help="untimed barrier-to-barrier regions of K steps between the warm-up and the timed repetitions (declared in the JSON line)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=20.0, help="total budget of the CPU baseline legs")
    p.add_argument("--no-sweep", action="store_true", help="skip the env-step kernel sweep over 4k..4M envs per launch (< 1 s)")
    p.add_argument("--sample-launch", action="store_true",
                   help="draw the minibatch with a launch of its own (hx_sample_batch) instead of inside learn()'s first launch (A/B of the fused draw)")
    p.add_argument("--sweep", action="store_true", help="(default) kept for older command lines")
    p.add_argument("--actions", default="policy", choices=["policy", "uniform"],
                   help="uniform: U(-1,1)^4 actions instead of the live actor (SURVEY.md 8d C2's second run, decoupled from the policy)")
    p.add_argument("--overlap", action="store_true",
                   help="issue the next act + env.step on a second stream beside critic-only learns (bit-identical results); measured on one "
                        "GPU it does not pay (every cross-stream event hand-off costs ~10 us on this runtime), so the default, at any N, is "
                        "the reference's strict act -> step -> sample -> learn order on one stream")
    p.add_argument("--front", dest="front", action="store_true", default=None,
                   help="(default where it applies and every rank has a GPU of its own: HIRL in fp32 or --dtype bf16 with the policy's actions in one "
                        "launch, batch <= 256) the FRONT launch (HirlEngine.step_learn, include/hirl4ucav.h hx_hirl_front): env step + the first two "
                        "launches of learn() as ONE launch; the minibatch is then drawn from the ring as it stood before this step's insert, without "
                        "the slots it may overwrite.  Given explicitly it is also taken by ranks that share a GPU (soak tests)")
    p.add_argument("--front-acting", dest="front_acting", default="x9", choices=["x9", "mfma"],
                   help="--dtype f32 in the front loop: x9 (default, engine.front_x9) = the acting workgroups' 256 -> 512 product through the exact "
                        "three-way bf16 split of both operands; mfma = fp32 MFMA")
    p.add_argument("--no-front", dest="front", action="store_false",
                   help="the reference's order on every step: act -> env step -> insert -> draw -> learn, each launch after the other (the minibatch "
                        "sees this step's transitions)")
    p.add_argument("--serial", action="store_true", help="(default) one stream")
    p.add_argument("--separate-launches", dest="separate_launches", action="store_true",
                   help="act and env step as two launches on every step (default: one fused launch, hx_actor_act_step)")
    p.add_argument("--staged", action="store_true",
                   help="run EXACTLY the launch sequence of a sharded rank on one rank too (stage entry points, split actor message, hx_adam_mixed, "
                        "both exchange calls — through torch.distributed when launched by torch.distributed.run): the cost of the N > 1 step")
    p.add_argument("--exchange", default="rccl", choices=["rccl", "rccl-torch", "rccl-bf16", "oneshot", "twostage", "twostage-bf16", "auto"],
                   help="gradient exchange of the sharded step: rccl = ncclAllReduce enqueued by the library on the engine's stream (hx_rccl_allreduce; "
                        "with the gloo test backend it falls back to torch.distributed); rccl-torch = torch.distributed.all_reduce; rccl-bf16 = the same "
                        "direct path with the messages cast to bf16 (half the wire bytes, bf16 sums); EXPERIMENTAL, never run on two physical GPUs: "
                        "oneshot (every rank reads every peer's message over hipIpc mappings), twostage (reduce-scatter + all-gather over the same "
                        "mappings), twostage-bf16 (its reduced slices as bf16); auto = whichever of rccl / twostage the probe (below) found faster")
    p.add_argument("--no-exchange-probe", dest="exchange_probe", action="store_false",
                   help="N > 1: skip the probe that times --probe-messages all-reduces of the critic's 1.1 MB message through rccl AND twostage on the "
                        "run's own process group before the timed loop (`exchange_probe` in the JSON line)")
    p.add_argument("--probe-messages", dest="probe_messages", type=int, default=200)
    p.add_argument("--b0-episodes", dest="b0_episodes", type=int, default=0,
                   help="B0 (reference plumbing) in SURVEY.md 8(d)'s form: this many episodes of 1,500 steps (3 = ~4 minutes on the GPU box's host); "
                        "0 (default): a few-second sample, so that the default run stays within minutes")
    p.add_argument("--exchange-timeout-ms", dest="exchange_timeout_ms", type=int, default=5000,
                   help="one-shot exchange: how long a rank waits for a peer's message before it raises (ranks that SHARE a GPU - tests - "
                        "only make progress through pre-emption and need far longer than ranks with a GPU each)")
    p.add_argument("--measure-steps", dest="measure_steps", type=int, default=256, help="steps of the instrumented second pass")
    p.add_argument("--inject-front-trip", dest="inject_front_trip", action="store_true",
                   help="tests: the LAST rank sets its front launch's status word after the timed region, as a wait that gave up would — every rank "
                        "must then repeat the run in the reference's order and rank 0 print a valid line")
    return p.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no launcher environment starts the N ranks itself
# ---------------------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def visible_gpus():
    import torch  # counting devices does not initialise the GPU

    return torch.cuda.device_count()


def launch_ranks(args, argv):
    n = visible_gpus()
    if n < args.gpus and os.environ.get("HX_BENCH_BACKEND", "nccl") == "nccl":
        sys.stderr.write(f"bench.py: {args.gpus} GPUs requested, {n} visible - refusing to print a {n}-GPU number as a {args.gpus}-GPU one\n")
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)  # a CHILD process: this one has not touched a GPU and simply relays the exit code


# ---------------------------------------------------------------------------------------------------------------------
# which loop
# ---------------------------------------------------------------------------------------------------------------------
SHARED_GPU = "ranks share a GPU"


def loop_choice(args, world, shared_device):
    """-> (front, reason).  The front launch (HirlEngine.step_learn / SacEngine.step_learn) where it applies; never by default when ranks share
    a device: its in-launch waits are argued for ONE process per GPU (include/hirl4ucav.h hx_hirl_front — the waiters of several processes add up;
    GPUTEST_r05: eight front-loop ranks on one GPU tripped and took the suite with them)."""
    sac = args.agent == "sac"
    why = []
    if args.actions == "uniform":
        why.append("uniform actions: no policy launch to fuse")
    if args.separate_launches:
        why.append("--separate-launches")
    if args.overlap:
        why.append("--overlap (two streams)")
    if args.sample_launch:
        why.append("--sample-launch (the draw as a launch of its own)")
    if args.batch > 256:
        why.append("batch > 256 (the front launch holds at most 16 row tiles of the minibatch)")
    if sac:
        if args.envs <= 8192:
            why.append("the SAC front launch exists beyond 8,192 envs")
        if world > 1:
            why.append("the SAC front launch is single-GPU")
    else:
        if args.dtype not in ("f32", "f32x9", "bf16"):
            why.append(f"--dtype {args.dtype}: the front launch takes fp32 networks or the bf16 actor + critic path")
        if args.dtype in ("f32", "f32x9") and args.front_acting == "mfma" and args.envs > 8192:
            why.append("fp32-MFMA acting in the front launch: at most 8,192 envs per GPU")
    if args.front is True:
        if why:
            raise SystemExit("bench.py --front: " + "; ".join(why))
        return True, "--front given" + (f" ({SHARED_GPU}: beyond the shape the in-launch waits are argued for; a trip falls back)" if shared_device else "")
    if args.front is False:
        return False, "--no-front given"
    if shared_device:
        why.append(f"{world} {SHARED_GPU}: the front launch's in-launch waits are for one process per GPU")
    if why:
        return False, "; ".join(why)
    return True, "default where it applies (one process per GPU, policy actions in one launch, batch <= 256)"


class Loop:
    """act -> env step (+ fused insert) -> sample -> learn, everything resident on one GPU.  step() enqueues the hot path and nothing
    else; step_measured() is the same step with events around the stages and, on request, act / env step as two launches with the
    env launch stamped — it is used only AFTER the timed region."""

    def __init__(self, args, rank, world, device, shared_device=False, forced_reason=None):
        import torch

        from hirl4ucav_amd import _lib
        from hirl4ucav_amd.agents.engine import HirlEngine
        from hirl4ucav_amd.environments.batched import BatchedHarfangEnv
        from hirl4ucav_amd.utils.buffer import DeviceReplay

        self.torch, self.lib = torch, _lib
        self.args, self.rank, self.world = args, rank, world
        n = args.envs
        self.max_step = 1900 if args.scenario == "circular" else 1500  # train_all.py:159-183
        self.replay = DeviceReplay(max(1 << 20, 2 * n), device)
        scenario = args.scenario
        if scenario == "mixed":  # contiguous thirds: wavefronts never mix scenarios
            scenario = np.sort((np.arange(n) + rank * n) % 3).astype(np.int32)
        self.env = BatchedHarfangEnv(n, scenario=scenario, device=device, seed=0, max_step=self.max_step, auto_reset=True,
                                     random_reset=True, env_id0=rank * n, replay=self.replay)
        rng = np.random.default_rng(0)  # same networks and expert set on every rank (replicas)
        actor, critic, bc = init_params(rng)
        self.sac = args.agent == "sac"
        self.uniform = args.actions == "uniform"
        if self.sac:
            from hirl4ucav_amd.agents.sac_engine import SacEngine

            def seq(p, names, last):
                return {"0.weight": p[names[0] + ".weight"], "0.bias": p[names[0] + ".bias"], "2.weight": p[names[1] + ".weight"],
                        "2.bias": p[names[1] + ".bias"], "4.weight": last[0], "4.bias": last[1]}

            w8 = rng.uniform(-0.04, 0.04, (8, 512)).astype(np.float32)
            self.eng = SacEngine(batch=args.batch, device=device)
            self.eng.load_params(seq(actor, ("full1", "full2"), (w8, np.zeros(8, np.float32))),
                                 seq(critic, ("full1", "full2"), (critic["final1.weight"], critic["final1.bias"])),
                                 seq(critic, ("full3", "full4"), (critic["final2.weight"], critic["final2.bias"])))
        else:
            self.eng = HirlEngine(batch=args.batch, device=device)
            self.eng.load_params(actor, critic, bc)
            if args.staged:
                self.eng.staged = True
                self.eng.sharded_sequence = True
            self.eng.set_act_dtype({"f32": "f32", "f32x9": "f32x9"}.get(args.dtype, "bf16"))
            self.eng.set_update_dtype("bf16" if args.dtype == "bf16" else "f32")
            exchanging = world > 1 or args.staged
            if world > 1 and args.exchange in ("oneshot", "twostage", "twostage-bf16"):
                self.eng.use_oneshot_exchange(timeout_ms=args.exchange_timeout_ms, two_stage=args.exchange != "oneshot",
                                              bf16=args.exchange == "twostage-bf16")
            elif args.exchange in ("rccl", "rccl-bf16") and getattr(args, "pg_backend", None) == "nccl" and exchanging:
                # (one GPU per rank: RCCL refuses ranks that share a device — the gloo test backend keeps torch.distributed)
                self.eng.use_rccl_direct(bf16=args.exchange == "rccl-bf16")
        es, ea = synthetic_expert(rng)
        # BC table rows (s, a) and the expert replay ring labelled on the GPU (train_all.py:289-306)
        bc_rows = np.zeros((es.shape[0], 32), np.float32)
        bc_rows[:, 0:13], bc_rows[:, 13:17] = es, ea
        self.bc_table = torch.from_numpy(bc_rows).to(device)
        m = es.shape[0] - 1
        s_t, a_t, ns_t = (torch.from_numpy(x).to(device).contiguous() for x in (es[:-1], ea[:-1], es[1:]))
        r = torch.zeros(m, device=device)
        sc = torch.zeros(m, dtype=torch.int8, device=device)
        dn = torch.zeros(m, dtype=torch.uint8, device=device)
        _lib.call("hx_label_transitions", s_t.data_ptr(), a_t.data_ptr(), ns_t.data_ptr(), m, r.data_ptr(), sc.data_ptr(), dn.data_ptr(),
                  _lib.stream_ptr())
        self.expert = DeviceReplay(m + 10, device)
        self.expert.store_rows(torch.cat([s_t, a_t, ns_t, r[:, None], dn.float()[:, None]], 1), sc)
        self.expert_num = 0  # steady state of the 128 -> 0 decay (train_all.py:356-357)
        self.env.reset()
        self.t = 0
        self.actions = torch.zeros((n, 4), device=device)
        from hirl4ucav_amd.utils.pipeline import VectorStepPipeline
        self.pipe = VectorStepPipeline(device, overlap=args.overlap and not args.serial and not self.sac)
        self.separate = args.separate_launches
        self.rec = {"act": [], "env": [], "act+env": [], "learn": [], "front+back": []}
        self.krec, self.krec_fused = [], []
        # act + env step + replay insert as ONE launch at every size (hx_actor_act_step / hx_sac_act_step: up to 8,192 envs one 16- / 32-row
        # workgroup per row tile with the env step on its first wave, beyond that the persistent kernel of csrc/hx_actp.hip)
        self.fused = not (self.uniform or self.separate)
        if not self.sac and args.front_acting == "mfma":
            self.eng.front_x9 = False
        if forced_reason is not None:  # the in-process repeat after a tripped front loop
            self.front, self.loop_reason = False, forced_reason
        else:
            self.front, self.loop_reason = loop_choice(args, world, shared_device)

    def close(self):
        if hasattr(self.eng, "close"):
            self.eng.close()

    # ---- the hot path ------------------------------------------------------------------------------------------------
    def _act_env(self, timed=None, split=False, stamp=None):
        e, env = self.eng, self.env
        t = timed or (lambda name, fn: fn())
        if self.uniform or self.separate or split:
            if self.uniform:  # env.action_space.sample() for every env (train_all.py:272)
                t("act", lambda: self.actions.uniform_(-1.0, 1.0))
            elif self.sac:
                t("act", lambda: e.act(env.obs, seed=1, row0=env.env_id0, out=self.actions))  # SacAgent.explore
            else:
                t("act", lambda: e.act(env.obs, sigma=0.1, seed=1, row0=env.env_id0, out=self.actions))  # actionNoise 0.1, HIRL.py:160
            if stamp is not None:
                env.time_next_steps(*stamp)
                self.krec.append(stamp)  # (recorded HERE: with --overlap the pipeline may skip this issue function altogether)
            t("env", lambda: env.step(self.actions))
            if stamp is not None:
                env.time_next_steps(None, None)
        else:
            if stamp is not None:  # the launch's own begin / end (hipExtLaunchKernelGGL events) of the fused act + env kernel
                env.time_next_steps(*stamp)
                self.krec_fused.append(stamp)
            if self.sac:   # explore + env.step in one launch
                t("act+env", lambda: e.act_step(env, seed=1, out=self.actions))
            else:          # chooseAction + env.step in one launch (same results, bit for bit: tests/test_hirl_gpu.py)
                t("act+env", lambda: e.act_step(env, sigma=0.1, seed=1, out=self.actions))
            if stamp is not None:
                env.time_next_steps(None, None)

    def _learn(self, act_env):
        e = self.eng
        defer = not (self.args.sample_launch or self.args.overlap)  # --overlap: the next env step may run beside learn(): draw first
        if self.sac:  # train_sac.py:401-403
            e.sample(self.replay, seed=2 + self.rank, defer=defer)
            e.learn()
            return
        # the draw and the gather ride in the first launch of learn() (hx_hirl_learn_sampled): same minibatch, one launch less
        e.sample(self.replay, self.expert, self.bc_table, n_main=e.batch - self.expert_num, seed=2 + self.rank, defer=defer)
        # a critic-only learn() leaves the acting network alone: the next act + env.step go out on the side stream now
        self.pipe.arm(act_env, acting_net_untouched=not e.actor_trainable, engine=e)
        w = self._bc_weight()
        # sharded path: the side stream is released at the gradient all-reduce; one-call path: right away
        if not e.staged:
            self.pipe.fire()
        e.learn(bc_weight_now=w, bc_warm_up_weight=0.0, before_exchange=self.pipe.fire)

    def _bc_weight(self):
        # soft weight: estimated at the start of every max_step-long "episode" of vector steps, kept in between
        # (the reference re-estimates at most once per episode)
        kind = self.args.type
        if kind == "soft":
            return 100 if (self.t % self.max_step == 0) else None
        if kind == "linear":  # bc_weight - episode / 5000, floored at 0 (train_all.py:328-331); episode = max_step vector steps
            return max(self.args.bc_weight - (self.t // self.max_step) / 5000.0, 0.0)
        return self.args.bc_weight

    def _front_step(self):
        """act + env step + replay insert AND launches A, B of learn() in one launch, then the rest of learn() (HirlEngine.step_learn)"""
        e = self.eng
        if self.sac:  # explore + env step + insert AND the first forward launch of learn() in one launch, then the rest (SacEngine.step_learn)
            e.step_learn(self.env, act_seed=1, out=self.actions, sample_seed=2 + self.rank)
            return
        e.step_learn(self.env, self.expert, self.bc_table, n_main=e.batch - self.expert_num, act_sigma=0.1, act_seed=1, out=self.actions,
                     sample_seed=2 + self.rank, bc_weight_now=self._bc_weight(), bc_warm_up_weight=0.0)

    def step(self):
        if self.front:
            self._front_step()
            self.t += 1
            return
        self.pipe.act_and_step(self._act_env)
        self._learn(self._act_env)
        self.pipe.join()
        self.t += 1

    # ---- the same step with instruments (second pass only) -----------------------------------------------------------------
    def step_measured(self, split, pool, kpool, stamp_front=True):
        def timed(name, fn):
            a, b = pool.pop(), pool.pop()
            a.record()
            fn()
            b.record()
            self.rec[name].append((a, b))

        stamp = None
        if not self.uniform and len(kpool) >= 2 and (split or not self.separate) and (split or stamp_front or not self.front):
            stamp = (kpool.pop(), kpool.pop())  # split: the env launch, else the fused act + env launch (filed by _act_env when it runs)
        if self.front and not split:  # the front launch stamped with its own begin / end; the whole step under one pair of stream events
            if stamp is not None:
                self.env.time_next_steps(*stamp)
                self.krec_fused.append(stamp)
            timed("front+back", self._front_step)
            self.env.time_next_steps(None, None)
            self.t += 1
            return
        act_env = lambda: self._act_env(timed, split, stamp)  # noqa: E731
        self.pipe.act_and_step(act_env)
        timed("learn", lambda: self._learn(act_env))
        self.pipe.join()
        self.t += 1

    def front_status(self):
        """the front launch's sticky status word (HIRL engine; the SAC front launch has no in-launch wait); synchronises"""
        return int(self.eng.front_status()) if (self.front and hasattr(self.eng, "front_status")) else 0


def workload_label(args):
    """what THIS run computes, from its arguments; a BASELINE.json configs[] index only where the arguments match that config"""
    if args.agent == "sac":
        what = f"{args.envs} parallel {args.scenario} envs per GPU, SAC fp32, 1 learn(B={args.batch}) per vector step"
        tag = " (BASELINE.json configs[2])" if (args.envs == 16384 and args.scenario == "serpentine" and args.batch == 128) else ""
        return what + tag
    dt = {"f32": "fp32", "bf16": "bf16 actor/critic (fp32 accumulate, fp32 master weights / Adam / LayerNorm) + fp32 dynamics",
          "bf16_policy": "bf16 policy inference (fp32 accumulate) + fp32 dynamics / update",
          "f32x9": "fp32 (acting kernel: the 256->512 product through the exact three-way bf16 split of both operands on bf16 MFMA at every size, "
                   "fp32 accumulate)"}[args.dtype]
    kind = f"HIRL-{args.type}" + (f" (bc_weight {args.bc_weight})" if args.type != "soft" else "")
    what = f"{args.envs} parallel {args.scenario} envs per GPU, {kind} {dt}, 1 learn(B={args.batch}) per vector step"
    tag = ""
    if args.actions == "policy" and args.batch == 128:
        if args.envs == 4096 and args.scenario == "straight_line" and args.type == "soft" and args.dtype == "f32":
            tag = " (BASELINE.json configs[1])"
        elif args.envs == 8192 and args.scenario == "circular" and args.type == "linear" and args.dtype == "f32":
            tag = " (one GPU's shard of BASELINE.json configs[3])"
        elif args.envs == 16384 and args.scenario == "mixed" and args.dtype == "bf16":
            tag = " (one GPU's shard of BASELINE.json configs[4])"
    return what + tag


# ---------------------------------------------------------------------------------------------------------------------
# the run of one rank
# ---------------------------------------------------------------------------------------------------------------------
class Ctx:
    """what every phase of a rank's run needs"""

    def __init__(self, args, rank, world, device, pg, backend, shared_device):
        self.args, self.rank, self.world, self.device, self.pg, self.backend, self.shared_device = args, rank, world, device, pg, backend, shared_device

    def barrier(self):
        import torch

        if self.world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, v, op="MAX"):
        import torch

        if self.world <= 1:
            return v
        t = torch.tensor([float(v)], dtype=torch.float64, device=self.device if self.backend == "nccl" else "cpu")
        torch.distributed.all_reduce(t, op=getattr(torch.distributed.ReduceOp, op))
        return float(t.item())


def settle(ctx, loop):
    """untimed steps of the same loop for --settle-s seconds; every rank leaves after the SAME number of collective calls"""
    import torch

    args, steps = ctx.args, 0
    if args.settle_s <= 0:
        return 0
    t_end = time.perf_counter() + args.settle_s
    while True:
        for _ in range(32):
            loop.step()
        steps += 32
        if ctx.world > 1:
            # the all-reduced flag alone decides (a rank-local clock test here could let one rank fall out of the loop while its peers enqueue
            # 32 more steps and one more flag exchange: mismatched collective sequences)
            flag = torch.tensor([1.0 if time.perf_counter() < t_end else 0.0], device=ctx.device)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            if float(flag.item()) == 0.0:
                break
        elif time.perf_counter() >= t_end:
            break
    torch.cuda.synchronize()
    return steps


def timed_regions(ctx, loop):
    """[--dry-regions untimed regions] + R repetitions of { barrier + synchronize -> EXACTLY K x step() -> synchronize + barrier }, max over ranks.
    The first barrier-to-barrier region behind the settle phase reads 2-15 us per step high in the 20-step form (69.4 / 56.8 / 54.5 us in a round-5
    run: the median of three then lands on the second-worst): hence the declared dry region."""
    args = ctx.args
    for _ in range(max(int(args.dry_regions), 0)):
        ctx.barrier()
        for _ in range(args.steps):
            loop.step()
        ctx.barrier()
    reps = []
    for _ in range(max(int(args.reps), 1)):
        ctx.barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loop.step()
        ctx.barrier()
        reps.append(ctx.max_over_ranks(time.perf_counter() - t0))
    return reps


def second_pass(ctx, loop):
    """stage events; the act + env (or front) launch stamped; every 4th step act and env step as two launches, the env launch stamped; events around
    every gradient exchange.  -> dict of medians and lists in us"""
    import torch

    args = ctx.args
    m_steps = max(int(args.measure_steps), 16)
    pool = [torch.cuda.Event(enable_timing=True) for _ in range(12 * (m_steps + 1))]
    L = loop.lib.load()
    kpool = [ctypes.c_void_p(L.hx_event_create()) for _ in range(2 * (m_steps + 2))]
    ar_events = []
    exchanging = ctx.world > 1 or (args.staged and ctx.pg and args.agent == "hirl")
    inner = getattr(loop.eng, "_allreduce", None)
    if exchanging:  # events around every gradient exchange (on the stream it is enqueued on)
        def timed_allreduce(t, kind=None):
            call = (lambda: inner(t, kind)) if kind is not None else (lambda: inner(t))
            if len(pool) >= 2:
                a, b = pool.pop(), pool.pop()
                a.record()
                out = call()
                b.record()
                ar_events.append((t.numel() * 4, a, b))
            else:
                out = call()
            return t if out is None else out
        loop.eng._allreduce = timed_allreduce
    for k in range(m_steps):
        # front loop: the front launch right behind a split step is not stamped (it follows a foreign env step: no pre-drawn minibatch, a draw
        # launch of its own in front of it, colder caches — a third of the stamped launches would be that slower first one)
        loop.step_measured(split=(k % 4 == 3), pool=pool, kpool=kpool, stamp_front=(k % 4 != 0))
    ctx.barrier()
    if exchanging:
        loop.eng._allreduce = inner
        if getattr(loop.eng, "xchg", None) is not None:
            loop.eng.xchg.check()  # a timed-out wait leaves garbage behind: fail loudly instead of printing a number
    out = {"m_steps": m_steps, "exchanging": exchanging,
           "med": {k: (float(np.median([a.elapsed_time(b) * 1e3 for a, b in v])) if v else None) for k, v in loop.rec.items()}}
    kern = RL.stamped_us(loop.lib, loop.krec)
    if not kern:  # uniform actions: the loop has no act launch to split off; stamp plain env steps
        kern = RL.stamped_env_us(loop.env, loop.actions, 32)
    out["env_kernel_us"] = kern
    out["fused_us"] = RL.stamped_us(loop.lib, loop.krec_fused) if loop.fused else []
    by = {}
    for nbytes, a, b in ar_events:
        by.setdefault(nbytes, []).append(a.elapsed_time(b) * 1e3)
    out["allreduce"] = by
    return out


def measure(ctx, loop):
    m = {"settle_steps": settle(ctx, loop)}
    for _ in range(ctx.args.warmup):
        loop.step()
    m["reps"] = timed_regions(ctx, loop)
    if ctx.args.inject_front_trip and loop.front and ctx.rank == ctx.world - 1 and getattr(loop.eng, "_front", None) is not None:
        loop.eng._front[1].fill_(1)  # as a launch-B workgroup that gave up would leave it
    m.update(second_pass(ctx, loop))
    m["front_status"] = int(ctx.max_over_ranks(loop.front_status()))  # ONE decision for all ranks
    return m


def reference_order_leg(ctx, args):
    """the SAME workload with every launch in the reference's order (act -> env step -> insert -> draw -> learn), timed the same way in the same
    process: what the front launch buys, and the figure to quote if the draw must see the current step's transitions"""
    ref_args = copy.copy(args)
    ref_args.front = False
    ref_loop = Loop(ref_args, ctx.rank, ctx.world, ctx.device)
    for _ in range(max(args.warmup, 64)):
        ref_loop.step()
    reps = timed_regions(ctx, ref_loop)
    dt = float(np.median(reps))
    n_total = args.envs * ctx.world
    return {"loop": "reference order (--no-front): the minibatch is drawn after this step's insert", "value": round(n_total * args.steps / dt, 1),
            "unit": "env steps/s", "ms_per_step": round(dt / args.steps * 1e3, 5), "update_steps_per_s": round(args.steps / dt, 1),
            "repetitions": {"count": len(reps), "statistic": "median", "ms_per_step": [round(t / args.steps * 1e3, 5) for t in reps]}}


def probe_exchanges(ctx, loop):
    """N > 1, before the timed loop: --probe-messages all-reduces of the critic's message (276,488 floats = 1.1 MB, the larger message of a sharded
    update) through BOTH transports — `rccl` in this process, as the run uses it (direct, or torch.distributed with the gloo test backend), with events
    on the stream it is enqueued on; the two-stage peer-read kernel over hipIpc mappings in CHILD processes (tools/exchange_probe.py, one per rank, a
    process group of their own): that kernel has never met two physical GPUs, and whatever its first contact with real xGMI does — a refused mapping,
    a wait that times out, a fault that aborts the process — must cost this run nothing but the probe's entry.  Every step ends with ONE all-rank
    agreement, so no rank is left alone in a collective.  -> the `exchange_probe` record (rank 0's view; every rank returns the same `fastest`)"""
    import torch

    args, dev = ctx.args, ctx.device
    count = max(int(args.probe_messages), 4) if not ctx.shared_device else min(max(int(args.probe_messages), 4), 24)
    rec = {"message_bytes": 4 * CRITIC_MESSAGE_FLOATS, "messages": count, "transports": []}

    def agree(ok):
        return ctx.max_over_ranks(0.0 if ok else 1.0) == 0.0

    # ---- rccl, as the run uses it ----
    msg = torch.ones(CRITIC_MESSAGE_FLOATS, dtype=torch.float32, device=dev)
    direct = getattr(loop.eng, "rccl", None)
    entry = {"transport": getattr(loop.eng, "exchange_name", "rccl") if direct is not None else f"torch.distributed ({ctx.backend})"}

    def rccl_one():
        if direct is not None:
            direct.allreduce(msg)
        else:
            torch.distributed.all_reduce(msg)

    ok, why = True, ""
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(count)]
    try:
        for _ in range(4):
            msg.fill_(1.0)
            rccl_one()
        ctx.barrier()
        for a, b in evs:
            msg.fill_(1.0)  # (outside the timed pair; keeps the values bounded over hundreds of sums)
            a.record()
            rccl_one()
            b.record()
        torch.cuda.synchronize()
        ok = bool((msg == float(ctx.world)).all())
        why = "" if ok else "the sum of ones is not the world size"
    except Exception as e:  # noqa: BLE001
        ok, why = False, f"{type(e).__name__}: {e}"
    if agree(ok):
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in evs)
        med = ctx.max_over_ranks(us[len(us) // 2])
        entry.update({"ok": True, "median_us": round(med, 2), "p10_us": round(us[len(us) // 10], 2), "p90_us": round(us[(9 * len(us)) // 10], 2),
                      "busbw_GBps": round(2 * (ctx.world - 1) / ctx.world * rec["message_bytes"] / med / 1e3, 2),
                      "statistic": "the slowest rank's median (p10 / p90: rank 0's)"})
    else:
        entry.update({"ok": False, "error": why or "another rank failed"})
    rec["transports"].append(entry)

    # ---- twostage, in child processes ----
    box = [free_port() if ctx.rank == 0 else None]
    torch.distributed.broadcast_object_list(box, src=0)
    env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC_")}  # (the children rendezvous on a store of their OWN)
    env.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(box[0]), "RANK": str(ctx.rank), "WORLD_SIZE": str(ctx.world),
                "LOCAL_RANK": str(ctx.device.index), "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")})
    cmd = [sys.executable, os.path.join(REPO, "tools", "exchange_probe.py"), "--floats", str(CRITIC_MESSAGE_FLOATS), "--messages", str(count),
           "--kind", "twostage", "--timeout-ms", str(args.exchange_timeout_ms)]
    entry, out, err, rc = {"transport": "twostage", "where": "child processes (tools/exchange_probe.py)"}, "", "", None
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)  # a CHILD: nothing here replaces this process
    try:
        out, err = child.communicate(timeout=90)
        rc = child.returncode
    except subprocess.TimeoutExpired:
        child.kill()
        out, err = child.communicate()
        rc = "timeout"
    if agree(rc == 0):
        line = [ln for ln in out.splitlines() if ln.startswith("{")]
        got = json.loads(line[-1]) if line else {}  # (rank 0's child prints the record; the other ranks only learn that it worked)
        keys = ("ok", "median_us", "p10_us", "p90_us", "busbw_GBps", "per_rank_median_us", "distinct_gpus", "statistic")
        entry.update({k: got.get(k) for k in keys} if got else {"ok": True})
    else:
        tail = " | ".join(err.strip().splitlines()[-3:])[-600:]
        entry.update({"ok": False, "error": f"rank {ctx.rank}: child exit {rc}" + (f": {tail}" if rc != 0 else " (another rank's child failed)")})
    rec["transports"].append(entry)
    # every rank must pick the same transport: rank 0 holds the children's figure, so rank 0 decides and broadcasts
    good = [t for t in rec["transports"] if t.get("ok") and t.get("median_us")]
    box = [min(good, key=lambda t: t["median_us"])["transport"] if (ctx.rank == 0 and good) else None]
    torch.distributed.broadcast_object_list(box, src=0)
    rec["fastest"] = box[0]
    return rec


def run_rank(args):
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    launched = "WORLD_SIZE" in os.environ and "RANK" in os.environ
    backend = os.environ.get("HX_BENCH_BACKEND", "nccl")  # gloo exists only to exercise this code path where all ranks share one GPU
    ngpu = torch.cuda.device_count()
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if ngpu < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if world > ngpu and backend == "nccl":
        raise SystemExit(f"bench.py: {world} GPUs requested, {ngpu} visible")
    local = int(os.environ.get("LOCAL_RANK", "0")) % ngpu
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    # Under a launcher the process group exists at ANY world size (RCCL = backend "nccl" on ROCm): with --staged a single rank then sends
    # its two messages per actor call through the collective library too.
    pg = world > 1 or launched
    own_pg = False
    if pg and not torch.distributed.is_initialized():  # (tools/bench_many.py runs several argument sets on ONE process group)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
        own_pg = True
    args.pg_backend = backend if pg else None
    ctx = Ctx(args, rank, world, device, pg, backend, shared_device=world > ngpu)

    probe = None
    if world > 1 and args.agent == "hirl" and (args.exchange == "auto" or (args.exchange == "rccl" and args.exchange_probe and not ctx.shared_device)):
        # (ranks that share a GPU — the gloo rehearsals — probe only on request: --exchange auto)
        # the probe needs the run's rccl transport: build the loop the run would use with `rccl`, probe, then pick
        chosen = args.exchange
        args.exchange = "rccl"
        loop = Loop(args, rank, world, device, shared_device=ctx.shared_device)
        probe = probe_exchanges(ctx, loop)
        if chosen == "auto" and probe["fastest"] == "twostage":
            loop.close()
            args.exchange = "twostage"
            loop = Loop(args, rank, world, device, shared_device=ctx.shared_device)
        probe["requested"], probe["chosen"] = chosen, args.exchange
    else:
        loop = Loop(args, rank, world, device, shared_device=ctx.shared_device)

    m = measure(ctx, loop)
    tripped = m["front_status"] if loop.front else 0
    if tripped:
        # an in-launch wait gave up somewhere: the minibatches read since are suspect and so is the number.  Repeat the run in the reference's order
        # (no in-launch waits) in THIS process — never a re-exec: this process has touched the GPU — and say so in the line.
        sys.stderr.write(f"bench.py rank {rank}: front launch tripped (status word {tripped}); repeating the run in the reference's order\n")
        loop.close()
        ref_args = copy.copy(args)
        ref_args.front, ref_args.inject_front_trip = False, False
        why = f"the front loop tripped (status word {tripped}: an in-launch wait gave up) and the run was repeated in the reference's order"
        loop = Loop(ref_args, rank, world, device, shared_device=ctx.shared_device, forced_reason=why)
        m = measure(ctx, loop)

    res = record(ctx, loop, m, tripped, probe)
    if rank == 0:
        if not args.no_sweep:
            res["roofline_env_sweep"] = RL.env_sweep(device)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baselines(args, device)
    loop.close()
    if pg and own_pg:
        torch.distributed.destroy_process_group()
        # RCCL writes a version banner through C stdio, which a redirected stdout holds back until exit: let it out first, so that the
        ctypes.CDLL(None).fflush(None)  # record is the LAST line of stdout
    return res if rank == 0 else None


def record(ctx, loop, m, tripped, probe):
    """the JSON line of this run (every rank takes part in the collectives; rank 0 prints)"""
    import torch

    args, world, rank, device = ctx.args, ctx.world, ctx.rank, ctx.device
    reps, med = m["reps"], m["med"]
    dt = float(np.median(reps))
    n_total = args.envs * world
    act_us, learn_us = med["act"], med["learn"]
    dtype_label = {"f32": "f32", "bf16": "bf16", "bf16_policy": "bf16 policy / f32 update",
                   "f32x9": "f32 (policy product: exact three-way bf16 split, six partial products)"}[args.dtype]
    peer = world > 1 and args.exchange in ("oneshot", "twostage", "twostage-bf16")
    res = {
        "metric": "env steps/sec (whole node) + HIRL update steps/sec at 4096 envs/GPU", "value": round(n_total * args.steps / dt, 1),
        "unit": "env steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_s": args.settle_s,
        "settle_steps": m["settle_steps"], "dry_regions": max(int(args.dry_regions), 0),
        "ms_per_step": round(dt / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype_label, "data": "synthetic",
        "repetitions": {"count": len(reps), "statistic": "median", "ms_per_step": [round(t / args.steps * 1e3, 5) for t in reps],
                        "value": [round(n_total * args.steps / t, 1) for t in reps]},
        "config": {"workload": workload_label(args), "envs_per_gpu": args.envs, "batch": args.batch, "actions": args.actions,
                   "act_env": "one launch (hx_actor_act_step / hx_sac_act_step)" if loop.fused else "two launches",
                   "issue_order": "two streams" if loop.pipe.overlap else "serial",
                   "update_path": "staged (the sharded rank's launch sequence)" if (args.staged or world > 1) and args.agent == "hirl" else "one-call",
                   "parallelism": f"dp{world}: env shards + replicated nets, {'peer-read ' + args.exchange if peer else 'RCCL'} "
                                  f"all-reduce of the flat gradients; effective batch = {args.batch} x {world}"},
        "update_steps_per_s": round(args.steps / dt, 1), "update_samples_per_s": round(args.batch * args.steps / dt, 1),
        "timed_region": "R x [K x step() between two barrier + synchronize pairs]; no events, no stamped or split launches inside (those are the "
                        "second pass); `dry_regions` untimed regions of the same shape run before the first timed one",
        "stage_us": {"pass": f"second pass, {m['m_steps']} steps after the timed region (events add a few us per step)",
                     "act+env_step(1 kernel)": None if med["act+env"] is None else round(med["act+env"], 2),
                     "act(own launch, every 4th step)": None if act_us is None else round(act_us, 2),
                     "env_step(own launch, every 4th step)": None if med["env"] is None else round(med["env"], 2),
                     "sample+learn": None if learn_us is None else round(learn_us, 2)},
    }
    cfg = res["config"]
    cfg["loop"] = "front" if loop.front else ("reference order (front tripped)" if tripped else "reference order")
    cfg["loop_reason"] = loop.loop_reason
    res["front_status"] = int(tripped)
    if loop.front and world == 1 and not ctx.pg:
        res["reference_order"] = reference_order_leg(ctx, args)
    if loop.front and args.agent == "hirl" and args.dtype == "f32" and loop.eng.front_x9:
        cfg["acting_product"] = ("fp32 operands, the 256 -> 512 product through the exact three-way bf16 split of both operands (hi | mid | lo), six of "
                                 "the nine partial products — the three below fp32 resolution are not formed — on bf16 MFMA with fp32 accumulation (the "
                                 "engine's fp32 acting format wherever it is the faster one: from 4,096 rows on, and in the front launch); max error vs "
                                 "fp64 2.7e-7 against 3.8e-7 for fp32 MFMA (16,384 rows, profiles/r05_x9_terms_ab.txt)")
    if loop.front:
        cfg["act_env"] = ("FRONT launch (hx_sac_front): explore + env step + replay insert + the first forward launch of learn() in one launch"
                          if args.agent == "sac" else
                          "FRONT launch (hx_hirl_front): act + env step + replay insert + launches A and B of learn() in one launch")
        cfg["draw"] = ("uniform without replacement over the transitions that are in the ring before AND after this step's insert (drawn from the ring "
                       "as it stood before the step, without the n slots the step may overwrite: HxSample.guard); --no-front draws after the insert, "
                       "like the reference")
        res["stage_us"]["front launch + rest of learn() (3 of every 4 steps of the second pass; the front launch stamped: + ~25 us of instruments)"] = \
            None if med["front+back"] is None else round(med["front+back"], 2)

    env_roof = RL.env_kernel_roof(args.envs, m["env_kernel_us"])
    if m["fused_us"]:
        res["roofline"] = RL.launch_roofline(args, loop, m["fused_us"])
        res["roofline_env_kernel"] = env_roof
    else:
        res["roofline"] = env_roof
    if learn_us:
        res["roofline_update"] = RL.update_roof(args, learn_us)
    if act_us and not loop.uniform:  # (with --actions uniform the 'act' stage is a torch uniform_ fill, not the policy)
        res["roofline_act"] = RL.act_roof(args, loop, act_us)
    if ctx.pg:
        ids = [None] * world
        torch.distributed.all_gather_object(ids, (socket.gethostname(), str(getattr(torch.cuda.get_device_properties(device.index), "uuid", device.index))))
        name = getattr(loop.eng, "exchange_name", "rccl")
        res["rccl_ranks"] = {"world_size": torch.distributed.get_world_size(), "backend": name if name.startswith("rccl-direct") else ctx.backend,
                             "process_group_backend": ctx.backend, "distinct_gpus": len(set(ids)), "exchange": name,
                             "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if ctx.backend == "nccl" else None}
    if probe is not None:
        res["exchange_probe"] = probe
    if world > 1:  # the replicas must still be bit-identical after every sharded update so far
        mine = torch.tensor([loop.eng.replica_checksum()], dtype=torch.int64, device=device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)
        res["replicas_identical"] = bool(all(int(c.item()) == int(mine.item()) for c in every))
    if m["allreduce"]:  # SURVEY.md 8d "collective bytes": per message size, median us and bus bandwidth 2 (n-1)/n * bytes / t
        res["allreduce"] = [{"bytes": k, "calls": len(v), "median_us": round(float(np.median(v)), 2),
                             "busbw_GBps": round(2 * (world - 1) / world * k / float(np.median(v)) / 1e3, 2)} for k, v in sorted(m["allreduce"].items())]
        if world == 1:
            res["allreduce_note"] = ("world size 1: the collective library short-circuits an in-place all-reduce of one rank (no kernel is launched); "
                                     "the figure is the host-side call on the stream, NOT an exchange time — the N > 1 term stays unmeasured on this box")
    res["env_stats"] = loop.env.stats_dict()
    del rank
    return res


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    launched = "WORLD_SIZE" in os.environ or "RANK" in os.environ
    if args.gpus > 1 and not launched:
        return launch_ranks(args, argv)
    rank = os.environ.get("RANK", "0")
    try:
        res = run_rank(args)
    except SystemExit as e:
        if isinstance(e.code, str):  # ONE greppable line per failing rank (torchrun's failure table drowns a traceback: GPUTEST_r05)
            sys.stderr.write(f"bench.py rank {rank}: SystemExit: {' '.join(e.code.split())}\n")
        raise
    except BaseException as e:  # noqa: BLE001
        traceback.print_exc()
        sys.stderr.write(f"bench.py rank {rank}: {type(e).__name__}: {' '.join(str(e).split())[:1500]}\n")
        sys.stderr.flush()
        return 1
    if res is not None:
        print(json.dumps(res), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
