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

Shape of a run: [settle: untimed steps for --settle-s seconds, so that clocks and caches are where a long run keeps them] ->
W warm-up steps -> [--dry-regions (1) untimed region of the same shape, declared] -> R = --reps (3) repetitions of { barrier + synchronize -> EXACTLY K steps with nothing but the hot path on the stream
-> synchronize + barrier (max over ranks) }; `value` / `ms_per_step` are the MEDIAN repetition, all R are listed.
Everything that needs events or stamped launches (stage times, the act + env launch's own duration for the roofline, the stand-alone env
kernel, all-reduce times) runs in a SECOND pass after the clock has been read.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP32_MATRIX_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 dense peak, same guide
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, same guide
ENV_BYTES_FUSED = 550            # algorithmic bytes per env-step with the fused replay insert
ENV_BYTES_PLAIN = 370
ACTOR_FLOP = 272896              # forward FLOPs per sample (2 * MAC, GEMMs only), SURVEY.md 8d
LEARN_FLOP_PER_SAMPLE = 3810816  # HIRL-soft learn(), averaged over the actor-every-2nd alternation, SURVEY.md 8d


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20000)  # ~2 s of GPU time
    p.add_argument("--warmup", type=int, default=2000)
    p.add_argument("--settle-s", dest="settle_s", type=float, default=1.5,
                   help="untimed steps of the same loop for this many seconds BEFORE the warm-up (clocks, caches, allocator); 0 = off")
    p.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    p.add_argument("--batch", type=int, default=128)
    p.add_argument("--scenario", default="straight_line", choices=["straight_line", "serpentine", "circular", "mixed"],
                   help="mixed: scenario id = env id mod 3, sorted so that each third of the shard is one scenario (BASELINE.json configs[4])")
    p.add_argument("--type", default="soft", choices=["soft", "linear", "fixed"], help="HIRL BC-weight schedule (train_all.py:328-339)")
    p.add_argument("--bc_weight", type=float, default=0.5, help="linear / fixed: the weight (configs[3]: linear, 0.5)")
    p.add_argument("--agent", default="hirl", choices=["hirl", "sac"], help="sac: BASELINE.json configs[2] (use --envs 16384 --scenario serpentine)")
    p.add_argument("--dtype", default="f32", choices=["f32", "bf16", "bf16_policy", "f32x9"],
                   help="f32x9: fp32 everywhere, the policy's 256->512 product of the ACTING kernel through the exact three-way bf16 split of both operands on the bf16 matrix cores "
                        "(fp32 operands, exact partial products, fp32 accumulation: fp32 results up to summation order; NOT the default); "
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
                   help="(default where it applies: HIRL in fp32 or --dtype bf16 with the policy's actions in one launch, batch <= 256) the FRONT launch "
                        "(HirlEngine.step_learn, include/hirl4ucav.h hx_hirl_front): env step + the first two launches of learn() as ONE launch; the minibatch is "
                        "then drawn from the ring as it stood before this step's insert, without the slots it may overwrite")
    p.add_argument("--front-acting", dest="front_acting", default="x9", choices=["x9", "mfma"],
                   help="--dtype f32 in the front loop: x9 (default, engine.front_x9) = the acting workgroups' 256 -> 512 product through the exact three-way bf16 split of both operands; mfma = fp32 MFMA")
    p.add_argument("--no-front", dest="front", action="store_false",
                   help="the reference's order on every step: act -> env step -> insert -> draw -> learn, each launch after the other (the minibatch sees this step's transitions)")
    p.add_argument("--serial", action="store_true", help="(default) one stream")
    p.add_argument("--separate-launches", dest="separate_launches", action="store_true",
                   help="act and env step as two launches on every step (default: one fused launch, hx_actor_act_step)")
    p.add_argument("--staged", action="store_true",
                   help="run EXACTLY the launch sequence of a sharded rank on one rank too (stage entry points, split actor message, hx_adam_mixed, "
                        "both exchange calls — through torch.distributed when launched by torch.distributed.run): the cost of the N > 1 step")
    p.add_argument("--exchange", default="rccl", choices=["rccl", "rccl-torch", "oneshot", "twostage", "twostage-bf16"],
                   help="gradient exchange of the sharded step: rccl = ncclAllReduce enqueued by the library on the engine's stream (hx_rccl_allreduce; "
                        "with the gloo test backend it falls back to torch.distributed); rccl-torch = torch.distributed.all_reduce; EXPERIMENTAL, never "
                        "run on two physical GPUs: oneshot (every rank reads every peer's message over hipIpc mappings), twostage (reduce-scatter + "
                        "all-gather over the same mappings), twostage-bf16 (its reduced slices as bf16)")
    p.add_argument("--b0-episodes", dest="b0_episodes", type=int, default=0,
                   help="B0 (reference plumbing) in SURVEY.md 8(d)'s form: this many episodes of 1,500 steps (3 = ~4 minutes on the GPU box's host); "
                        "0 (default): a few-second sample, so that the default run stays within minutes")
    p.add_argument("--exchange-timeout-ms", dest="exchange_timeout_ms", type=int, default=5000,
                   help="one-shot exchange: how long a rank waits for a peer's message before it raises (ranks that SHARE a GPU - tests - "
                        "only make progress through pre-emption and need far longer than ranks with a GPU each)")
    p.add_argument("--measure-steps", dest="measure_steps", type=int, default=256, help="steps of the instrumented second pass")
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
# synthetic inputs
# ---------------------------------------------------------------------------------------------------------------------
def synthetic_expert(rng, n=20000):
    """states U(-1,1)^13 with cols 7,8 in {+-1}, col 12 in [0, 0.2]; actions U(-1,1)^3 ++ fire +-1, P(+1) = 1e-3."""
    s = rng.uniform(-1, 1, (n, 13)).astype(np.float32)
    s[:, 7] = np.where(rng.random(n) < 0.5, 1, -1)
    s[:, 8] = np.where(rng.random(n) < 0.5, 1, -1)
    s[:, 12] = rng.uniform(0, 0.2, n)
    a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    a[:, 3] = np.where(rng.random(n) < 1e-3, 1, -1)
    return s, a


def init_params(rng):
    """Seeded-init networks with the reference's bounds (HIRL.py:26-37,111-121)."""
    import math

    def U(b, shape):
        return rng.uniform(-b, b, shape).astype(np.float32)

    def block(in_dim, out_dim, names):
        fa, la, fb, lb, fin = names
        return {fa + ".weight": U(math.sqrt(6 / in_dim), (256, in_dim)), fa + ".bias": U(1 / math.sqrt(in_dim), (256,)),
                la + ".weight": np.ones(256, np.float32), la + ".bias": np.zeros(256, np.float32),
                fb + ".weight": U(math.sqrt(6 / 256), (512, 256)), fb + ".bias": U(1 / 16, (512,)),
                lb + ".weight": np.ones(512, np.float32), lb + ".bias": np.zeros(512, np.float32),
                fin + ".weight": U(1 / math.sqrt(512), (out_dim, 512)), fin + ".bias": U(1 / math.sqrt(512), (out_dim,))}

    actor = block(13, 4, ("full1", "layernorm1", "full2", "layernorm2", "final"))
    bc = block(13, 4, ("full1", "layernorm1", "full2", "layernorm2", "final"))
    critic = block(17, 1, ("full1", "layernorm1", "full2", "layernorm2", "final1"))
    critic.update(block(17, 1, ("full3", "layernorm3", "full4", "layernorm4", "final2")))
    return actor, critic, bc


class Loop:
    """act -> env step (+ fused insert) -> sample -> learn, everything resident on one GPU.  step() enqueues the hot path and nothing
    else; step_measured() is the same step with events around the stages and, on request, act / env step as two launches with the
    env launch stamped — it is used only AFTER the timed region."""

    def __init__(self, args, rank, world, device):
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
            if world > 1 and args.exchange in ("oneshot", "twostage", "twostage-bf16"):
                self.eng.use_oneshot_exchange(timeout_ms=args.exchange_timeout_ms, two_stage=args.exchange != "oneshot", bf16=args.exchange == "twostage-bf16")
            elif args.exchange == "rccl" and getattr(args, "pg_backend", None) == "nccl" and (world > 1 or args.staged):
                self.eng.use_rccl_direct()  # (one GPU per rank: RCCL refuses ranks that share a device — the gloo test backend keeps torch.distributed)
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
        if not self.sac and getattr(args, "front_acting", "x9") == "mfma":
            self.eng.front_x9 = False
        sac_front_ok = self.sac and n > 8192 and self.world == 1 and not (self.uniform or self.separate or args.overlap or args.sample_launch) and args.batch <= 256
        front_ok = sac_front_ok or not (self.sac or self.uniform or self.separate or args.overlap or args.sample_launch or args.dtype not in ("f32", "f32x9", "bf16") or n > ((1 << 30) if (args.dtype == "bf16" or getattr(args, "front_acting", "x9") == "x9") else 8192) or args.batch > 256)
        if getattr(args, "front", None) and not front_ok:
            raise SystemExit("--front: HIRL in fp32 or bf16 (actor and critic), policy actions in one launch, one stream, at most 8,192 envs per GPU with fp32-MFMA acting (exact-split acting and bf16: any number) and batch 256")
        self.front = front_ok if getattr(args, "front", None) is None else bool(args.front)

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
        if self.sac:  # train_sac.py:401-403
            e.sample(self.replay, seed=2 + self.rank, defer=not (self.args.sample_launch or self.args.overlap))
            e.learn()
            return
        # the draw and the gather ride in the first launch of learn() (hx_hirl_learn_sampled): same minibatch, one launch less
        e.sample(self.replay, self.expert, self.bc_table, n_main=e.batch - self.expert_num, seed=2 + self.rank, defer=not (self.args.sample_launch or self.args.overlap))  # --overlap: the next env step may run beside learn(): draw first
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
        if self.sac:  # explore + env step + insert AND the first forward launch of learn() in one launch, then the rest of learn() (SacEngine.step_learn)
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
        torch = self.torch

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
        del torch


# ---------------------------------------------------------------------------------------------------------------------
# CPU baselines (BASELINE.md 3 / SURVEY.md 8d): reported beside the GPU number, never a target.  The oracle may serve this leg.
# ---------------------------------------------------------------------------------------------------------------------
def host_cpu():
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return model, os.cpu_count() or 1


def best_torch_threads(step, cores):
    """torch's intra-op thread count that makes `step()` (ONE WHOLE vector step of the loop: actor forward for all envs, the env step on its worker
    threads, one learn at B = 128 — calibrated on the torch part alone, a box once chose 64 threads that then ran the loop 5x slower beside the 16 env
    threads) fastest on this host: the
    default (one thread per physical core: 128 on the GPU boxes) is thread-oversubscribed for a 128-row MLP and reads 3-10x too slow — the CPU
    figure is reported at its best, not at its worst.  -> (threads, {threads: seconds per step})"""
    import torch

    tried = {}
    for t in [c for c in (1, 2, 4, 8, 16, 32, 64) if c <= cores] or [1]:
        torch.set_num_threads(t)
        step()  # warm
        t0 = time.perf_counter()
        for _ in range(4):
            step()
        tried[t] = round((time.perf_counter() - t0) / 4, 5)
    best = min(tried, key=tried.get)
    torch.set_num_threads(best)
    return best, tried


def baseline_port(args, seconds):
    """The oracle timed on a BOUNDED sample of the same workload: the same loop (actor forward for all envs, env step for all
    envs with insert — the envs split over host threads, >= 256 envs each —, one HIRL learn at B = 128) for as many vector steps as fit."""
    import torch
    from concurrent.futures import ThreadPoolExecutor

    from oracle import hirl_oracle as H
    from tests import _oracle as ox

    n = args.envs
    cores = os.cpu_count() or 1
    rng = np.random.default_rng(0)
    actor, critic, bc = init_params(rng)
    es, ea = synthetic_expert(rng)
    o = H.HirlOracle(actor, critic, bc)
    envs, obs = ox.reset_batch(n, 0, 1, seed=0)
    workers = max(1, min(cores, n // 256))
    chunks = [(k * n // workers, (k + 1) * n // workers) for k in range(workers)]
    cap = 1 << 14
    rings = [np.zeros((cap, 32), np.float32) for _ in chunks]   # one private ring segment per worker (no shared head on the CPU side)
    totals = [np.zeros(1, np.uint64) for _ in chunks]
    epi = np.zeros(n, np.uint32)
    pool = ThreadPoolExecutor(len(chunks))  # ctypes releases the GIL inside ox_env_step_batch

    def work(k, a):
        lo, hi = chunks[k]
        ox.step_batch(envs[lo:hi], a[lo:hi], obs[lo:hi], max_step=1500, auto_reset=1, randomize=1, seed=0, env_id0=lo, episode_ctr=epi[lo:hi],
                      ring=rings[k], total=totals[k])

    count = [0]

    def loop_step():
        k = count[0]
        a = o.choose_action(obs, rng.normal(0, 0.1, (n, 4)).astype(np.float32))
        list(pool.map(lambda j: work(j, a), range(len(chunks))))
        ring = rings[k % len(rings)]
        m = max(min(int(totals[k % len(rings)][0]), cap), 1)
        rows = ring[rng.integers(0, m, args.batch)]
        ibc = rng.integers(0, es.shape[0], args.batch)
        o.learn((rows[:, 0:13], rows[:, 13:17], rows[:, 17:30], rows[:, 30], rows[:, 31]), (es[ibc], ea[ibc]),
                rng.normal(0, 0.2, 4).astype(np.float32), 100 if k == 0 else o.bc_weight, 0.0)
        count[0] = k + 1

    torch_threads, tried = best_torch_threads(loop_step, cores)
    steps, t0 = 0, time.perf_counter()
    while True:
        loop_step()
        steps += 1
        dt = time.perf_counter() - t0
        if dt > seconds or steps >= 2000:
            break
    pool.shutdown()
    # `cores` = the threads this baseline actually USED at once (the env phase and the torch phase alternate: the larger of the two), not the box's 256
    return {"value": round(n * steps / dt, 1), "unit": "env steps/s", "cores": max(len(chunks), int(torch_threads)), "kind": "port",
            "sample": f"{steps} vector steps of {n} envs in {dt:.1f} s: oracle C env step on {len(chunks)} threads + torch-CPU actor forward and "
                      f"HIRL learn on {torch_threads} threads (the fastest of {sorted(tried)}: seconds per vector step {tried})",
            "threads": {"env_step": len(chunks), "torch": torch_threads, "torch_tried_s_per_step": tried, "logical_cores": cores},
            "update_steps_per_s": round(steps / dt, 2)}


def baseline_port_sac(args, seconds):
    """The SAC loop (train_sac.py:238-241,401-403) on the oracle: SacOracle.explore for all envs, the oracle's C env step with insert on host
    threads, one SacOracle.learn at B = 128 per vector step — a BOUNDED sample of the same workload."""
    import torch
    from concurrent.futures import ThreadPoolExecutor

    from oracle import sac_oracle as S
    from tests import _oracle as ox

    n = args.envs
    cores = os.cpu_count() or 1
    rng = np.random.default_rng(0)
    o = S.SacOracle(S.init_mlp(rng, 13, 8), S.init_mlp(rng, 17, 1), S.init_mlp(rng, 17, 1))
    scen = {"straight_line": 0, "serpentine": 1, "circular": 2}.get(args.scenario, 0)
    envs, obs = ox.reset_batch(n, scen, 1, seed=0)
    workers = max(1, min(cores, n // 256))
    chunks = [(k * n // workers, (k + 1) * n // workers) for k in range(workers)]
    cap = 1 << 14
    rings = [np.zeros((cap, 32), np.float32) for _ in chunks]
    totals = [np.zeros(1, np.uint64) for _ in chunks]
    epi = np.zeros(n, np.uint32)
    pool = ThreadPoolExecutor(len(chunks))

    def work(k, a):
        lo, hi = chunks[k]
        ox.step_batch(envs[lo:hi], a[lo:hi], obs[lo:hi], max_step=1500, auto_reset=1, randomize=1, seed=0, env_id0=lo, episode_ctr=epi[lo:hi],
                      ring=rings[k], total=totals[k])

    count = [0]

    def loop_step():
        k = count[0]
        a = o.explore(obs, rng.normal(0, 1, (n, 4)).astype(np.float32)).astype(np.float32)
        list(pool.map(lambda j: work(j, a), range(len(chunks))))
        ring = rings[k % len(rings)]
        m = max(min(int(totals[k % len(rings)][0]), cap), 1)
        rows = ring[rng.integers(0, m, args.batch)]
        o.learn((rows[:, 0:13], rows[:, 13:17], rows[:, 30], rows[:, 17:30], rows[:, 31]), rng.normal(0, 1, (args.batch, 4)).astype(np.float32),
                rng.normal(0, 1, (args.batch, 4)).astype(np.float32))
        count[0] = k + 1

    torch_threads, tried = best_torch_threads(loop_step, cores)
    steps, t0 = 0, time.perf_counter()
    while True:
        loop_step()
        steps += 1
        dt = time.perf_counter() - t0
        if dt > seconds or steps >= 2000:
            break
    pool.shutdown()
    return {"value": round(n * steps / dt, 1), "unit": "env steps/s", "cores": max(len(chunks), int(torch_threads)), "kind": "port",
            "sample": f"{steps} vector steps of {n} envs in {dt:.1f} s: oracle C env step on {len(chunks)} threads + torch-CPU SAC explore and "
                      f"learn on {torch_threads} threads (the fastest of {sorted(tried)}: seconds per vector step {tried})",
            "threads": {"env_step": len(chunks), "torch": torch_threads, "torch_tried_s_per_step": tried, "logical_cores": cores},
            "update_steps_per_s": round(steps / dt, 2)}


def baseline_batched_cpu(seconds):
    """B1: the batched CPU integrator alone on ALL host cores — the oracle's C env step (no policy, no update), one thread per logical
    core, 1,024 envs per thread, uniform random actions: the fairest CPU line for the env half of the metric."""
    from concurrent.futures import ThreadPoolExecutor

    from tests import _oracle as ox

    cores = os.cpu_count() or 1
    per = 1024
    n = cores * per
    envs, obs = ox.reset_batch(n, 0, 1, seed=0)
    rng = np.random.default_rng(1)
    a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    epi = np.zeros(n, np.uint32)
    pool = ThreadPoolExecutor(cores)

    def work(k):
        lo, hi = k * per, (k + 1) * per
        for _ in range(8):  # 8 steps per dispatch: the Python hand-off stays below 1 % of the thread's time
            ox.step_batch(envs[lo:hi], a[lo:hi], obs[lo:hi], max_step=1500, auto_reset=1, randomize=1, seed=0, env_id0=lo, episode_ctr=epi[lo:hi])

    list(pool.map(work, range(cores)))  # warm
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        list(pool.map(work, range(cores)))
        steps += 8
    dt = time.perf_counter() - t0
    pool.shutdown()
    return {"value": round(n * steps / dt, 1), "unit": "env steps/s", "cores": cores, "kind": "port",
            "what": "the oracle's batched env step alone (scalar C, -O2, one thread per logical core, no policy / update)",
            "sample": f"{steps} vector steps of {n} envs in {dt:.1f} s"}


def baseline_reference_plumbing(seconds, episodes=0, episode_steps=1500):
    """episodes > 0: SURVEY.md 8(d)'s form of B0 — that many episodes of `episode_steps` steps (straight_line's maxStep), each opened with
    random_reset's message sequence (HarfangEnv_GYM.py:51-81), however long it takes (--b0-episodes: minutes; the default run takes a few-second sample).
    B0 (configs[0]): ONE env behind the reference's loopback framing — 4-byte big-endian length + JSON (socket_lib.py:86-143), the
    wrapper's message sequence per step (HarfangEnv_GYM.py:139-158: 6 level setters [+ FIRE_MISSILE] + UPDATE_SCENE; :193-251: 4
    request/reply read-backs), no TCP_NODELAY on the client (the reference sets none) — with the oracle simulator as the server and
    the oracle's eager CPU HIRL agent doing chooseAction + learn every step, as train_all.py:341-361 does."""
    import torch

    from hirl4ucav_amd.environments.wire import ALLY, OPPO, WireServer
    from oracle import hirl_oracle as H
    from tests._wire_backend import OracleSimBackend

    rng = np.random.default_rng(0)
    actor, critic, bc = init_params(rng)
    es, ea = synthetic_expert(rng)
    agent = H.HirlOracle(actor, critic, bc)
    srv = WireServer(OracleSimBackend(), "127.0.0.1", 0).start()
    sock = socket.create_connection(("127.0.0.1", srv.port))
    sent = [0]

    def send(command, **a):
        body = json.dumps({"command": command, "args": a}).encode()
        sock.sendall(len(body).to_bytes(4, "big") + body)
        sent[0] += 1

    def exact(k):
        buf = b""
        while len(buf) < k:
            buf += sock.recv(k - len(buf))
        return buf

    def ask(command, **a):
        send(command, **a)
        return json.loads(exact(int.from_bytes(exact(4), "big")).decode())

    def observe():
        pa, po = ask("GET_PLANE_STATE", plane_id=ALLY), ask("GET_PLANE_STATE", plane_id=OPPO)
        h = ask("GET_HEALTH", machine_id=OPPO)["health_level"]
        slot = ask("GET_MISSILESDEVICE_SLOTS_STATE", machine_id=ALLY)["missiles_slots"][0]
        d = (np.asarray(pa["position"]) - np.asarray(po["position"])) / 10000.0
        return np.concatenate([d, np.asarray(pa["Euler_angles"]) / np.pi, [pa["target_angle"] / 180.0, 1.0 if pa["target_locked"] else -1.0,
                               1.0 if slot else -1.0], np.asarray(po["Euler_angles"]) / np.pi, [h]]), float(np.linalg.norm(d) * 10000.0)

    def random_reset():  # _random_reset_machine + _reset_missile + the first observation (HarfangEnv_GYM.py:51-81, :171-188)
        send("RESET_MACHINE", machine_id=ALLY)
        send("RESET_MACHINE", machine_id=OPPO)
        send("SET_HEALTH", machine_id=OPPO, health_level=0.2)
        send("RESET_MACHINE_MATRIX", machine_id=OPPO, position=[0, 4200, 0], rotation=[0, 0, 0])
        send("RESET_MACHINE_MATRIX", machine_id=ALLY, position=[int(rng.integers(-100, 101)), 3500 + int(rng.integers(-100, 101)), -4000 + int(rng.integers(-100, 101))],
             rotation=[0, 0, 0])
        send("SET_PLANE_THRUST", plane_id=ALLY, thrust_level=1.0)
        send("SET_PLANE_THRUST", plane_id=OPPO, thrust_level=0.6)
        send("SET_PLANE_LINEAR_SPEED", plane_id=ALLY, linear_speed=300.0)
        send("SET_PLANE_LINEAR_SPEED", plane_id=OPPO, linear_speed=200.0)
        send("REARM_MACHINE", machine_id=ALLY)
        return observe()[0]

    obs, _ = observe()
    # the replay memory starts with 128 rows, as after the reference's exploration episodes (train_all.py:266-282): learn() runs from step 1
    mem = [rng.uniform(-1, 1, 32).astype(np.float32) for _ in range(128)]
    steps, t0 = 0, time.perf_counter()
    limit = episodes * episode_steps if episodes > 0 else 5000
    while True:
        if episodes > 0 and steps % episode_steps == 0:
            obs = random_reset()
        a = agent.choose_action(obs.astype(np.float32)[None], rng.normal(0, 0.1, 4).astype(np.float32))[0]
        send("SET_PLANE_PITCH", plane_id=ALLY, pitch_level=float(a[0]))
        send("SET_PLANE_ROLL", plane_id=ALLY, roll_level=float(a[1]))
        send("SET_PLANE_YAW", plane_id=ALLY, yaw_level=float(a[2]))
        send("SET_PLANE_PITCH", plane_id=OPPO, pitch_level=0.0)
        send("SET_PLANE_ROLL", plane_id=OPPO, roll_level=0.0)
        send("SET_PLANE_YAW", plane_id=OPPO, yaw_level=0.0)
        if a[3] > 0:
            send("FIRE_MISSILE", machine_id=ALLY, slot_id=0)
        send("UPDATE_SCENE")
        nobs, dist = observe()
        r = -1e-4 * dist - 10.0 * nobs[6] - (8.0 if a[3] > 0 else 0.0)
        mem.append(np.concatenate([obs, a, nobs, [r, 0.0]]).astype(np.float32))
        obs = nobs
        if len(mem) >= 128:
            rows = np.stack([mem[i] for i in rng.choice(len(mem), 128, replace=False)])
            ibc = rng.choice(es.shape[0], 128, replace=False)
            agent.learn((rows[:, 0:13], rows[:, 13:17], rows[:, 17:30], rows[:, 30], rows[:, 31]), (es[ibc], ea[ibc]),
                        rng.normal(0, 0.2, 4).astype(np.float32), 100 if len(mem) == 128 else agent.bc_weight, 0.0)
        steps += 1
        dt = time.perf_counter() - t0
        if (episodes <= 0 and dt > seconds) or steps >= limit:
            break
        if len(mem) > 20000:  # (the sample is uniform over the memory: keep the full form's host memory bounded)
            del mem[:10000]
    sock.close()
    srv.close()
    return {"value": round(steps / dt, 2), "unit": "env steps/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "what": "configs[0]: 1 env behind the reference's socket framing (loopback TCP + JSON), eager CPU HIRL chooseAction + learn per step",
            "sample": (f"{episodes} episodes x {episode_steps} steps = " if episodes > 0 else "") +
                      f"{steps} env steps in {dt:.1f} s, {sent[0] / max(steps, 1):.1f} messages per step, one learn(B=128) per step"}


def baseline_eager_rocm_learn(args, seconds, device):
    """B2: the same HIRL learn() as stock eager PyTorch-ROCm ops on the GPU (the oracle's functional restatement with its tensors
    on the device) — what the reference's agent costs when only its device string changes."""
    import torch

    from oracle import hirl_oracle as H

    rng = np.random.default_rng(0)
    actor, critic, bc = init_params(rng)
    es, ea = synthetic_expert(rng)
    o = H.HirlOracle(actor, critic, bc, device=device)
    rows = torch.from_numpy(rng.uniform(-1, 1, (4096, 32)).astype(np.float32)).to(device)
    est, eat = torch.from_numpy(es).to(device), torch.from_numpy(ea).to(device)
    noise = torch.from_numpy(rng.normal(0, 0.2, 4).astype(np.float32)).to(device)

    def one(k):
        idx = torch.randint(0, rows.shape[0], (args.batch,), device=device)
        ib = torch.randint(0, est.shape[0], (args.batch,), device=device)
        b = rows[idx]
        o.learn((b[:, 0:13], b[:, 13:17], b[:, 17:30], b[:, 30], (b[:, 31] > 0.9).float()), (est[ib], eat[ib]), noise, 100 if k == 0 else o.bc_weight, 0.0)

    for k in range(4):
        one(k)
    torch.cuda.synchronize()
    steps, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds and steps < 4000:
        one(steps + 4)
        steps += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return {"value": round(steps / dt, 1), "unit": "learn() calls/s", "kind": "port",
            "what": "HIRL learn(B=128) as eager PyTorch-ROCm ops on the same GPU (autograd + hand-written Adam/Polyak of the oracle)",
            "sample": f"{steps} calls in {dt:.1f} s"}


# ---------------------------------------------------------------------------------------------------------------------
def stamped_env_us(env, actions, launches):
    """The env-step kernel's OWN duration (begin/end stamps of hipExtLaunchKernelGGL, what rocprofv3 reports) over `launches` launches."""
    import torch

    from hirl4ucav_amd import _lib

    L = _lib.load()
    evs = [(ctypes.c_void_p(L.hx_event_create()), ctypes.c_void_p(L.hx_event_create())) for _ in range(launches)]
    for s, e in evs:
        env.time_next_steps(s, e)
        env.step(actions)
    env.time_next_steps(None, None)
    torch.cuda.synchronize()
    us = []
    for s, e in evs:
        v = ctypes.c_float()
        _lib.call("hx_event_elapsed_us", s, e, ctypes.byref(v))
        us.append(v.value)
        L.hx_event_destroy(s)
        L.hx_event_destroy(e)
    return us


def env_sweep(device):
    """The env-step kernel with the fused insert over 4k..4M envs per launch: the kernel's own duration, algorithmic 550 B per env-step."""
    import torch

    from hirl4ucav_amd.environments.batched import BatchedHarfangEnv
    from hirl4ucav_amd.utils.buffer import DeviceReplay

    out = []
    for n in (4096, 65536, 1 << 20, 1 << 22):
        rep = DeviceReplay(max(2 * n, 1 << 20), device)
        env = BatchedHarfangEnv(n, scenario="straight_line", device=device, seed=0, max_step=1500, replay=rep)
        env.reset()
        a = torch.rand(n, 4, device=device) * 2 - 1
        for _ in range(3):
            env.step(a)
        us = float(np.median(stamped_env_us(env, a, 16)))
        out.append({"envs_per_launch": n, "us": round(us, 2), "GBps": round(ENV_BYTES_FUSED * n / us / 1e3, 1),
                    "frac": round(ENV_BYTES_FUSED * n / us / 1e3 / HBM_PEAK_GBPS, 4)})
        del env, rep
        torch.cuda.empty_cache()
    return out


def profile_traffic(envs):
    """HBM bytes per launch of the env-step kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per
    pass, gfx950 FETCH x2 calibration: tools/pmc_env.py).  Not measured by this run: reported under its own key with the file it came from."""
    f = os.path.join(REPO, "profiles", "pmc_env_traffic.json")
    if not os.path.exists(f):
        return None
    with open(f) as fh:
        doc = json.load(fh)
    pmc = doc.get(str(envs))
    if not pmc:
        return None
    return {"bytes": pmc["traffic_bytes"], "fetch_bytes": pmc["fetch_bytes"], "write_bytes": pmc["write_bytes"], "ratio_to_algorithmic": pmc["ratio"],
            "source": "profiles/pmc_env_traffic.json", "kernel_build": doc.get("kernel_build"), "measured_at_commit": doc.get("measured_at_commit"),
            "note": "separate rocprofv3 --pmc passes of tools/pmc_env.py at this size; a profile artefact, not a measurement of this run"}


POLICY_FLOP_SAC = 2 * (13 * 256 + 256 * 512 + 512 * 8)  # GaussianPolicy forward, GEMMs only


def workload_label(args):
    """what THIS run computes, from its arguments; a BASELINE.json configs[] index only where the arguments match that config"""
    if args.agent == "sac":
        what = f"{args.envs} parallel {args.scenario} envs per GPU, SAC fp32, 1 learn(B={args.batch}) per vector step"
        cfg = 2 if (args.envs == 16384 and args.scenario == "serpentine") else None
        tag = " (BASELINE.json configs[2])" if cfg == 2 else ""
        return what + tag
    dt = {"f32": "fp32", "bf16": "bf16 actor/critic (fp32 accumulate, fp32 master weights / Adam / LayerNorm) + fp32 dynamics",
          "bf16_policy": "bf16 policy inference (fp32 accumulate) + fp32 dynamics / update",
          "f32x9": "fp32 (acting kernel: the 256->512 product through the exact three-way bf16 split of both operands on bf16 MFMA, fp32 accumulate)"}[args.dtype]
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
    if pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.distributed.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
    args.pg_backend = backend if pg else None
    loop = Loop(args, rank, world, device)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # ---- settle (declared, untimed), warm-up ----
    settle_steps = 0
    if args.settle_s > 0:
        t_end = time.perf_counter() + args.settle_s
        while True:
            for _ in range(32):
                loop.step()
            settle_steps += 32
            if world > 1:
                # every rank leaves the phase after the SAME number of collective calls: the all-reduced flag alone decides (a rank-local
                # clock test here could let one rank fall out of the loop while its peers enqueue 32 more steps and one more flag exchange)
                flag = torch.tensor([1.0 if time.perf_counter() < t_end else 0.0], device=device)
                torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
                if float(flag.item()) == 0.0:
                    break
            elif time.perf_counter() >= t_end:
                break
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        loop.step()
    # one UNTIMED dry region of the same shape (declared: `dry_regions`): the first barrier-to-barrier region behind the settle phase reads 2-15 us per step
    # high in the 20-step form (69.4 / 56.8 / 54.5 and 61.4 / 53.3 / 54.1 us in two round-5 runs: the median of three then lands on the second-worst)
    for _ in range(max(int(args.dry_regions), 0)):
        barrier()
        for _ in range(args.steps):
            loop.step()
        barrier()
    # ---- the timed region: R repetitions of K steps, nothing else on the stream ----
    reps = []
    for _ in range(max(int(args.reps), 1)):
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loop.step()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=device)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt.item())
        reps.append(dt)
    dt = float(np.median(reps))

    # ---- second pass: stage events; the act + env launch stamped; every 4th step act and env step as two launches, the env launch stamped ----
    ar_events = []
    m_steps = max(int(args.measure_steps), 16)
    pool = [torch.cuda.Event(enable_timing=True) for _ in range(12 * (m_steps + 1))]
    L = loop.lib.load()
    kpool = [ctypes.c_void_p(L.hx_event_create()) for _ in range(2 * (m_steps + 2))]
    exchanging = world > 1 or (args.staged and pg and args.agent == "hirl")
    if exchanging:  # exchange step: events around every gradient exchange (on the stream it is enqueued on)
        inner = loop.eng._allreduce

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
        # front loop: the front launch right behind a split step is not stamped (it follows a foreign env step: no pre-drawn minibatch, a draw launch
        # of its own in front of it, colder caches — a third of the stamped launches would be that slower first one)
        loop.step_measured(split=(k % 4 == 3), pool=pool, kpool=kpool, stamp_front=(k % 4 != 0))
    barrier()
    if loop.front and hasattr(loop.eng, "front_check"):
        loop.eng.front_check()  # an in-launch wait that gave up leaves a minibatch half read: fail loudly instead of printing a number
    if exchanging:
        loop.eng._allreduce = inner
        if getattr(loop.eng, "xchg", None) is not None:
            loop.eng.xchg.check()  # a timed-out wait leaves garbage behind: fail loudly instead of printing a number
    med = {k: (float(np.median([a.elapsed_time(b) * 1e3 for a, b in v])) if v else None) for k, v in loop.rec.items()}

    def stamped_us(pairs):
        out = []
        for a, b in pairs:
            us = ctypes.c_float()
            loop.lib.call("hx_event_elapsed_us", a, b, ctypes.byref(us))
            out.append(us.value)
        return out

    kern = stamped_us(loop.krec)
    if not kern:  # uniform actions: the loop has no act launch to split off; stamp plain env steps
        kern = stamped_env_us(loop.env, loop.actions, 32)
    env_kernel_us = float(np.mean(kern))  # mean, like the rocprofv3 --stats average it must agree with
    fused = stamped_us(loop.krec_fused) if loop.fused else []
    act_us, learn_us = med["act"], med["learn"]

    n_total = args.envs * world
    value = n_total * args.steps / dt
    res = {
        "metric": "env steps/sec (whole node) + HIRL update steps/sec at 4096 envs/GPU", "value": round(value, 1),
        "unit": "env steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "settle_s": args.settle_s, "settle_steps": settle_steps, "dry_regions": max(int(args.dry_regions), 0),
        "ms_per_step": round(dt / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"f32": "f32", "bf16": "bf16", "bf16_policy": "bf16 policy / f32 update", "f32x9": "f32 (policy product: exact bf16 x 9 split)"}[args.dtype], "data": "synthetic",
        "repetitions": {"count": len(reps), "statistic": "median", "ms_per_step": [round(t / args.steps * 1e3, 5) for t in reps],
                        "value": [round(n_total * args.steps / t, 1) for t in reps]},
        "config": {"workload": workload_label(args), "envs_per_gpu": args.envs, "batch": args.batch,
                   "actions": args.actions, "act_env": "one launch (hx_actor_act_step / hx_sac_act_step)" if loop.fused else "two launches",
                   "issue_order": "two streams" if loop.pipe.overlap else "serial",
                   "update_path": "staged (the sharded rank's launch sequence)" if (args.staged or world > 1) and args.agent == "hirl" else "one-call",
                   "parallelism": f"dp{world}: env shards + replicated nets, {'peer-read ' + args.exchange if (world > 1 and args.exchange in ('oneshot', 'twostage', 'twostage-bf16')) else 'RCCL'} "
                                  f"all-reduce of the flat gradients; effective batch = {args.batch} x {world}"},
        "update_steps_per_s": round(args.steps / dt, 1),
        "timed_region": "R x [K x step() between two barrier + synchronize pairs]; no events, no stamped or split launches inside (those are the second pass); `dry_regions` untimed regions of the same shape run before the first timed one",
        "stage_us": {"pass": f"second pass, {m_steps} steps after the timed region (events add a few us per step)",
                     "act+env_step(1 kernel)": None if med["act+env"] is None else round(med["act+env"], 2),
                     "act(own launch, every 4th step)": None if act_us is None else round(act_us, 2),
                     "env_step(own launch, every 4th step)": None if med["env"] is None else round(med["env"], 2),
                     "sample+learn": None if learn_us is None else round(learn_us, 2)},
    }
    if loop.front and world == 1 and not pg:
        # the SAME workload with every launch in the reference's order (act -> env step -> insert -> draw -> learn), timed the same way in the same process:
        # what the front launch buys, and the figure to quote if the draw must see the current step's transitions
        import copy
        ref_args = copy.copy(args)
        ref_args.front = False
        ref_loop = Loop(ref_args, rank, world, device)
        for _ in range(max(args.warmup, 64)):
            ref_loop.step()
        for _ in range(max(int(args.dry_regions), 0)):  # (the same untimed dry region as the line's own loop)
            barrier()
            for _ in range(args.steps):
                ref_loop.step()
            barrier()
        ref_reps = []
        for _ in range(max(int(args.reps), 1)):
            barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ref_loop.step()
            barrier()
            ref_reps.append(time.perf_counter() - t0)
        ref_dt = float(np.median(ref_reps))
        res["reference_order"] = {"loop": "reference order (--no-front): the minibatch is drawn after this step's insert", "value": round(n_total * args.steps / ref_dt, 1),
                                  "unit": "env steps/s", "ms_per_step": round(ref_dt / args.steps * 1e3, 5), "update_steps_per_s": round(args.steps / ref_dt, 1),
                                  "repetitions": {"count": len(ref_reps), "statistic": "median", "ms_per_step": [round(t / args.steps * 1e3, 5) for t in ref_reps]}}
        del ref_loop
    res["config"]["loop"] = "front" if loop.front else "reference order"
    if loop.front and args.agent == "hirl" and args.dtype == "f32" and loop.eng.front_x9:
        res["config"]["acting_product"] = ("fp32 operands, the 256 -> 512 product through the exact three-way bf16 split of both operands (hi | mid | lo), six of the nine partial products — the three below fp32 resolution are not formed — on bf16 MFMA with fp32 accumulation (the engine's fp32 acting "
                                           "format wherever it is the faster one: from 4,096 rows on, and in the front launch); max error vs fp64 2.7e-7 against 3.8e-7 for fp32 MFMA (16,384 rows, profiles/r05_x9_terms_ab.txt)")
    if loop.front:
        res["config"]["act_env"] = ("FRONT launch (hx_sac_front): explore + env step + replay insert + the first forward launch of learn() in one launch" if args.agent == "sac" else
                                    "FRONT launch (hx_hirl_front): act + env step + replay insert + launches A and B of learn() in one launch")
        res["config"]["draw"] = ("uniform without replacement over the transitions that are in the ring before AND after this step's insert (drawn from the ring as it stood "
                                 "before the step, without the n slots the step may overwrite: HxSample.guard); --no-front draws after the insert, like the reference")
        res["stage_us"]["front launch + rest of learn() (3 of every 4 steps of the second pass; the front launch stamped: + ~25 us of instruments)"] = None if med["front+back"] is None else round(med["front+back"], 2)

    env_roof = {"kernel": "env_step_kernel<PAIR, INSERT, EPB> (hx_env.hip)", "bound": "hbm",
                "achieved": round(ENV_BYTES_FUSED * args.envs / env_kernel_us / 1e3, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(ENV_BYTES_FUSED * args.envs / env_kernel_us / 1e3 / HBM_PEAK_GBPS, 4), "traffic": (profile_traffic(args.envs) or {}).get("bytes"),
                "bytes_per_launch": ENV_BYTES_FUSED * args.envs, "us_per_launch": round(env_kernel_us, 2), "launches_timed": len(kern),
                "timing": "the kernel's own begin/end stamps (hipExtLaunchKernelGGL events) on the launch stream, mean over the launches of "
                          "the second pass in which act and env step are issued as two launches",
                "traffic_note": "HBM bytes per launch from the committed rocprofv3 --pmc passes of this kernel at this size (not collectable inside this "
                                "process): see traffic_from_profiles; null when no pass exists for the size",
                "traffic_from_profiles": profile_traffic(args.envs)}
    if fused:
        # the kernel the timed loop RUNS: policy inference + env step + replay insert in one launch.  Both roofs are quoted; `bound` names the nearer.
        us = float(np.mean(fused))
        fused_pmc = (profile_traffic(f"fused_{args.envs}") if args.dtype == "f32" else profile_traffic(f"fused_bf16_{args.envs}") if args.dtype == "bf16" else None) \
            if args.agent == "hirl" else None
        if loop.front:  # (the front launch has PMC passes of its own)
            fused_pmc = profile_traffic(f"front_{args.envs}") if args.dtype == "f32" else profile_traffic(f"front_bf16_{args.envs}") if args.dtype == "bf16" else None
        fp32_equiv = (POLICY_FLOP_SAC if args.agent == "sac" else ACTOR_FLOP) * args.envs
        # which matrix-core instruction the 256 -> 512 product runs on: fp32 MFMA; bf16 MFMA; or — the fp32 HIRL policy from 16,384 rows on, and
        # --dtype f32x9 at every size — SIX bf16 MFMAs per fp32 product (the exact hi | mid | lo split of both operands, its partial products above fp32 resolution): `roofline.executed` prices those against the bf16 peak
        x9 = args.agent == "hirl" and (args.dtype == "f32x9" or (args.dtype == "f32" and (args.envs >= 4096 or (loop.front and loop.eng.front_x9))))
        if args.dtype in ("f32", "f32x9") or args.agent == "sac":  # ALGORITHMIC FLOPs against the dense matrix peak of the dtype the path computes in
            flop, peak = fp32_equiv, FP32_MATRIX_PEAK_TFLOPS
        else:
            flop, peak = fp32_equiv, BF16_MATRIX_PEAK_TFLOPS
        if loop.front and args.agent != "sac":  # + the forward passes of launches A and B over the minibatch: 3 + 2 nets on a critic-only call, 4 + 4 on an actor call (every 2nd)
            flop += int(6.5 * args.batch * ACTOR_FLOP)
        # algorithmic bytes of the launch: 550 B per env step — and, for the front launch, the five networks of launches A and B read once
        # (target actor, critic x 2, target critic x 2: the figure tools/pmc_traffic_json.py sets the counter passes against; VERDICT r4: the flop side
        # already counted A and B, the byte side did not)
        nbytes = ENV_BYTES_FUSED * args.envs + (4 * (138756 + 4 * 138244) if (loop.front and args.agent != "sac") else 0)
        tf, gb = flop / us / 1e6, nbytes / us / 1e3
        mf, hf = tf / peak, gb / HBM_PEAK_GBPS
        hbm = {"bound": "hbm", "achieved": round(gb, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(hf, 4), "bytes_per_launch": nbytes}
        mfma = {"bound": "mfma", "achieved": round(tf, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(mf, 4), "flop_per_launch": flop}
        if x9:  # the fp32 product runs as 9 bf16 MFMAs per 32 k (exact split, fp32 accumulate): what the matrix cores EXECUTE, against their bf16 peak
            ex = flop + 5 * 2 * 256 * 512 * args.envs
            mfma["executed"] = {"what": "the acting workgroups' 256 -> 512 product as SIX bf16 MFMAs per fp32 product (exact hi | mid | lo split of both operands; the three partial products below fp32 resolution are not formed): executed FLOPs against the bf16 "
                                        "dense peak — the matrix cores' utilisation; `achieved` above is the fp32 arithmetic the launch delivers, which the split lets exceed "
                                        "what v_mfma_f32_16x16x4_f32 could (peak 157.3)",
                                "flop_per_launch": ex, "achieved": round(ex / us / 1e6, 2), "peak": BF16_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "frac": round(ex / us / 1e6 / BF16_MATRIX_PEAK_TFLOPS, 4)}
        first, second = (mfma, hbm) if mf >= hf else (hbm, mfma)
        persistent = args.envs > 8192
        front_name = None
        if loop.front and args.agent == "sac":
            front_name = ("actps_sac_front_kernel<MODE> (hx_front.hip): the persistent streaming acting workgroups (Gaussian policy + env step + fused replay insert) with the "
                          "first forward launch of learn() (policy(s'), policy(s), Q1/Q2(s, a)) behind them")
        elif loop.front:  # + the forward passes of launches A (3.5 nets on average) and B (3 on average) over the minibatch
            front_name = ("act_front_kernel<RELU, X3, BF16> (hx_front.hip): the acting workgroups (32 rows each: policy inference + env step + fused replay insert) on half "
                          "of the CUs, launches A and B of learn() (target actor, critics; target critics) on the other half") if args.envs <= (4096 if args.dtype == "bf16" else 8192) else \
                         ("actp_front_kernel<RELU> / actps_front_kernel<RELU> (hx_front.hip): persistent acting workgroups (bf16: weight-stationary, two thirds of the CUs; exact split: "
                          "one 64-row pass each) with the env step + fused replay insert in their tail, launches A and B of learn() on the CUs they leave")
        plain_name = (("act_persist_*_kernel<..., ENV = true> (hx_actp.hip): persistent workgroups (one per CU) looping over their row tiles, env step + "
                       "fused replay insert in the launch's tail") if persistent else
                      "act_fused_kernel<NRT, GAUSS, ENV = true, ...> (hx_act.hip): policy inference + env step + fused replay insert")
        res["roofline"] = {"kernel": ((front_name + "; FLOPs: the policy's over the envs + the 6.5 forward passes (average) of launches A and B over the minibatch") if front_name else plain_name) + ", the dominant kernel of the timed loop", **first, "traffic": (fused_pmc or {}).get("bytes"), "other_roof": second,
                           "us_per_launch": round(us, 2), "launches_timed": len(fused),
                           "timing": "the kernel's own begin/end stamps (hipExtLaunchKernelGGL events) on the launch stream, mean over the fused launches "
                                     "of the second pass (" + ("2 of every 4 steps: every 4th issues act, env step and learn() as separate launches, and the front launch behind it is not stamped" if loop.front else "3 of every 4 steps") + ")",
                           "note": ("bound by CU time: 128 acting workgroups of 32 rows beside 320-448 update workgroups on the other 128 CUs; neither roof is near (DESIGN.md section 4 K5)" if loop.front else
                                    "vector-issue / LDS bound tile loop (LayerNorm + head per row), DESIGN.md section 4" if persistent else
                                    "latency-bound at this size: 256 workgroups, one round; neither roof is near (DESIGN.md section 4)"),
                           "traffic_note": "HBM bytes per launch from the committed rocprofv3 --pmc passes of this launch (HIRL, this policy format, this size; FETCH_SIZE "
                                           "calibrated x2, WRITE_SIZE: tools/pmc_env_passes.sh); each of the 8 XCDs pulls the policy's weights into its own L2 once "
                                           "per launch, hence a few x the env's 550 B/env-step at small sizes; null where no pass exists",
                           "traffic_from_profiles": fused_pmc}
        res["roofline_env_kernel"] = env_roof
    else:
        res["roofline"] = env_roof
    if learn_us:
        peak_u = BF16_MATRIX_PEAK_TFLOPS if args.dtype == "bf16" else FP32_MATRIX_PEAK_TFLOPS
        res["roofline_update"] = {"kernels": "fwd_l2/bwd_l2/wgrad(+adam) (one learn, minibatch draw included)", "bound": "mfma", "unit": "TFLOP/s",
                                  "achieved": round(LEARN_FLOP_PER_SAMPLE * args.batch / learn_us / 1e6, 3), "peak": peak_u,
                                  "frac": round(LEARN_FLOP_PER_SAMPLE * args.batch / learn_us / 1e6 / peak_u, 5),
                                  "us_per_learn": round(learn_us, 2), "timing": "torch events around learn() in the second pass (median)"}
    if act_us and not loop.uniform:  # (with --actions uniform the 'act' stage is a torch uniform_ fill, not the policy)
        peak = FP32_MATRIX_PEAK_TFLOPS if (args.dtype in ("f32", "f32x9") or args.agent == "sac") else BF16_MATRIX_PEAK_TFLOPS
        flop = (POLICY_FLOP_SAC if args.agent == "sac" else ACTOR_FLOP) * args.envs
        res["roofline_act"] = {"kernels": "the acting kernel (ENV = false) as its own launch (every 4th step of the second pass); fp32-equivalent FLOPs", "bound": "mfma", "unit": "TFLOP/s",
                               "achieved": round(flop / act_us / 1e6, 3), "peak": peak,
                               "frac": round(flop / act_us / 1e6 / peak, 5), "us": round(act_us, 2), "timing": "torch events (median)"}
    if pg:
        ids = [None] * world
        torch.distributed.all_gather_object(ids, (socket.gethostname(), str(getattr(torch.cuda.get_device_properties(local), "uuid", local))))
        direct = getattr(loop.eng, "exchange_name", "") == "rccl-direct"
        res["rccl_ranks"] = {"world_size": torch.distributed.get_world_size(), "backend": "rccl-direct" if direct else backend, "process_group_backend": backend,
                             "distinct_gpus": len(set(ids)),
                             "exchange": getattr(loop.eng, "exchange_name", "rccl"),
                             "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None}
    if world > 1:  # the replicas must still be bit-identical after every sharded update so far
        mine = torch.tensor([loop.eng.replica_checksum()], dtype=torch.int64, device=device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(every, mine)
        res["replicas_identical"] = bool(all(int(c.item()) == int(mine.item()) for c in every))
    if ar_events:  # SURVEY.md 8d "collective bytes": per message size, median us and bus bandwidth 2 (n-1)/n * bytes / t
        by = {}
        for nbytes, a, b in ar_events:
            by.setdefault(nbytes, []).append(a.elapsed_time(b) * 1e3)
        res["allreduce"] = [{"bytes": k, "calls": len(v), "median_us": round(float(np.median(v)), 2),
                             "busbw_GBps": round(2 * (world - 1) / world * k / float(np.median(v)) / 1e3, 2)} for k, v in sorted(by.items())]
        if world == 1:
            res["allreduce_note"] = ("world size 1: the collective library short-circuits an in-place all-reduce of one rank (no kernel is launched); the "
                                     "figure is the host-side call on the stream, NOT an exchange time — the N > 1 term stays unmeasured on this box")
    res["env_stats"] = loop.env.stats_dict()
    if rank == 0:
        if not args.no_sweep:
            res["roofline_env_sweep"] = env_sweep(device)
        if world == 1 and not args.no_cpu_baseline:
            model, cores = host_cpu()
            budget = max(args.cpu_seconds, 1.0)
            if args.agent == "hirl":
                res["cpu_baseline"] = baseline_port(args, 0.4 * budget)
                res["cpu_baseline"]["host"] = f"{model}, {cores} logical cores"
                res["cpu_baseline"]["b0_reference_plumbing"] = baseline_reference_plumbing(0.3 * budget, episodes=args.b0_episodes)
                res["cpu_baseline"]["b1_batched_cpu"] = baseline_batched_cpu(0.15 * budget)
                res["cpu_baseline"]["b2_eager_rocm_learn"] = baseline_eager_rocm_learn(args, 0.15 * budget, device)
            else:
                res["cpu_baseline"] = baseline_port_sac(args, 0.7 * budget)
                res["cpu_baseline"]["host"] = f"{model}, {cores} logical cores"
                res["cpu_baseline"]["b1_batched_cpu"] = baseline_batched_cpu(0.3 * budget)
    if pg:
        torch.distributed.destroy_process_group()
        # RCCL writes a version banner through C stdio, which a redirected stdout holds back until exit: let it out first, so that the
        ctypes.CDLL(None).fflush(None)  # record is the LAST line of stdout
    if rank == 0:
        print(json.dumps(res), flush=True)
    return 0


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    launched = "WORLD_SIZE" in os.environ or "RANK" in os.environ
    if args.gpus > 1 and not launched:
        return launch_ranks(args, argv)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
