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

    python bench.py                       # 1 GPU, defaults finish in about a minute (incl. the bounded CPU baseline)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP32_MATRIX_PEAK_TFLOPS = 157.3  # v_mfma_f32_16x16x4_f32 dense peak, same guide
ENV_BYTES_FUSED = 550            # algorithmic bytes per env-step with the fused replay insert
ENV_BYTES_PLAIN = 370
ACTOR_FLOP = 272896              # forward FLOPs per sample (2 * MAC, GEMMs only), SURVEY.md 8d
LEARN_FLOP_PER_SAMPLE = 3810816  # HIRL-soft learn(), averaged over the actor-every-2nd alternation, SURVEY.md 8d


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20000)  # ~2.2 s of GPU time: long enough for the clocks to settle
    p.add_argument("--warmup", type=int, default=2000)
    p.add_argument("--envs", type=int, default=4096, help="envs per GPU")
    p.add_argument("--batch", type=int, default=128)
    p.add_argument("--scenario", default="straight_line", choices=["straight_line", "serpentine", "circular", "mixed"],
                   help="mixed: scenario id = env id mod 3, sorted so that each third of the shard is one scenario (BASELINE.json configs[4])")
    p.add_argument("--type", default="soft", choices=["soft", "linear", "fixed"], help="HIRL BC-weight schedule (train_all.py:328-339)")
    p.add_argument("--bc_weight", type=float, default=0.5, help="linear / fixed: the weight (configs[3]: linear, 0.5)")
    p.add_argument("--agent", default="hirl", choices=["hirl", "sac"], help="sac: BASELINE.json configs[2] (use --envs 16384 --scenario serpentine)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    p.add_argument("--sweep", action="store_true", help="also sweep the env-step kernel over 4k..4M envs per launch")
    p.add_argument("--actions", default="policy", choices=["policy", "uniform"],
                   help="uniform: U(-1,1)^4 actions instead of the live actor (SURVEY.md 8d C2's second run, decoupled from the policy)")
    p.add_argument("--overlap", action="store_true",
                   help="issue the next act + env.step on a second stream beside critic-only learns (bit-identical results); in the sharded "
                        "path the side stream is released at the gradient all-reduce. Measured on one GPU it does not pay — 107.8 vs 100.1 "
                        "us/step one-call, 117.5 vs 100.1 staged: every cross-stream event hand-off costs ~10 us on this runtime — so the "
                        "default, at any N, is the reference's strict act -> step -> sample -> learn order on one stream")
    p.add_argument("--serial", action="store_true", help="(default) one stream")
    p.add_argument("--separate-launches", dest="separate_launches", action="store_true",
                   help="act and env step as two launches on every step (default: one fused launch, hx_actor_act_step)")
    p.add_argument("--staged", action="store_true",
                   help="use the stage-by-stage update path of the sharded build on one rank too (costs of the N > 1 launch sequence)")
    return p.parse_args()


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
    """act -> env step (+ fused insert) -> sample -> learn, everything resident on one GPU."""

    def __init__(self, args, rank, world, device):
        from hirl4ucav_amd import _lib
        from hirl4ucav_amd.agents.engine import HirlEngine
        from hirl4ucav_amd.environments.batched import BatchedHarfangEnv
        from hirl4ucav_amd.utils.buffer import DeviceReplay

        self.lib = _lib
        self.args, self.rank, self.world = args, rank, world
        n = args.envs
        self.max_step = 1900 if args.scenario == "circular" else 1500  # train_all.py:159-183
        self.replay = DeviceReplay(max(1 << 20, 2 * n), device)
        scenario = getattr(args, "scenario", "straight_line")
        if scenario == "mixed":  # contiguous thirds: wavefronts never mix scenarios
            scenario = np.sort((np.arange(n) + rank * n) % 3).astype(np.int32)
        self.env = BatchedHarfangEnv(n, scenario=scenario, device=device, seed=0, max_step=self.max_step, auto_reset=True,
                                     random_reset=True, env_id0=rank * n, replay=self.replay)
        rng = np.random.default_rng(0)  # same networks and expert set on every rank (replicas)
        actor, critic, bc = init_params(rng)
        self.sac = getattr(args, "agent", "hirl") == "sac"
        self.uniform = getattr(args, "actions", "policy") == "uniform"
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
            if getattr(args, "staged", False):
                self.eng.staged = True
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
        self.expert_len, self.bc_len = m, es.shape[0]
        self.expert_num = 0  # steady state of the 128 -> 0 decay (train_all.py:356-357)
        self.env.reset()
        self.t = 0
        self.actions = torch.zeros((n, 4), device=device)
        from hirl4ucav_amd.utils.pipeline import VectorStepPipeline
        self.pipe = VectorStepPipeline(device, overlap=getattr(args, "overlap", False) and not getattr(args, "serial", False) and not self.sac)
        self.record, self.rec, self.pool = False, {"act": [], "env": [], "act+env": [], "learn": []}, []
        self.separate = getattr(args, "separate_launches", False)
        self.kpool, self.krec, self.act_env_calls = [], [], 0

    def _timed(self, name, fn):
        """Bracket fn() with HIP events on the stream it launches on (the current one) while recording is on."""
        if not self.record:
            return fn()
        if len(self.pool) < 2:  # events are made outside the timed region (creation stalls the issue thread); none left: untimed
            return fn()
        a, b = self.pool.pop(), self.pool.pop()
        a.record()
        fn()
        b.record()
        self.rec[name].append((a, b))

    def _act_env(self):
        e, env = self.eng, self.env
        # The env kernel's own duration (what rocprofv3 reports) is sampled on every 8th recorded step: there the two stages go
        # out as separate launches, act then a STAMPED env step (a stamped launch costs ~9 us extra, hence not on every step).
        self.act_env_calls += 1  # (not self.t: with two streams this function runs one step ahead, always on odd t)
        stamped = self.record and len(self.kpool) >= 2 and self.act_env_calls % 8 == 0
        if self.uniform or self.separate or stamped:
            if self.uniform:  # env.action_space.sample() for every env (train_all.py:272)
                self._timed("act", lambda: self.actions.uniform_(-1.0, 1.0))
            elif self.sac:
                self._timed("act", lambda: e.act(env.obs, seed=1, row0=env.env_id0, out=self.actions))  # SacAgent.explore
            else:
                self._timed("act", lambda: e.act(env.obs, sigma=0.1, seed=1, row0=env.env_id0, out=self.actions))  # actionNoise 0.1, HIRL.py:160
            if stamped:
                a, b = self.kpool.pop(), self.kpool.pop()
                env.time_next_steps(a, b)
                self.krec.append((a, b))
            self._timed("env", lambda: env.step(self.actions))
            env.time_next_steps(None, None)
        elif self.sac:   # explore + env.step in one launch
            self._timed("act+env", lambda: e.act_step(env, seed=1, out=self.actions))
        else:            # chooseAction + env.step in one launch (same results, bit for bit: tests/test_hirl_gpu.py)
            self._timed("act+env", lambda: e.act_step(env, sigma=0.1, seed=1, out=self.actions))

    def _learn(self):
        e = self.eng
        if self.sac:  # train_sac.py:401-403
            e.sample(self.replay, seed=2 + self.rank)
            e.learn()
            return
        e.sample(self.replay, self.expert, self.bc_table, n_main=e.batch - self.expert_num, seed=2 + self.rank)
        # a critic-only learn() leaves the acting network alone: the next act + env.step go out on the side stream now
        self.pipe.arm(self._act_env, acting_net_untouched=not e.actor_trainable)
        # soft weight: estimated at the start of every max_step-long "episode" of vector steps, kept in between
        # (the reference re-estimates at most once per episode)
        kind = getattr(self.args, "type", "soft")
        if kind == "soft":
            w = 100 if (self.t % self.max_step == 0) else None
        elif kind == "linear":  # bc_weight - episode / 5000, floored at 0 (train_all.py:328-331); episode = max_step vector steps
            w = max(self.args.bc_weight - (self.t // self.max_step) / 5000.0, 0.0)
        else:
            w = self.args.bc_weight
        # sharded path: the side stream is released at the gradient all-reduce; one-call path: right away
        if not e.staged:
            self.pipe.fire()
        e.learn(bc_weight_now=w, bc_warm_up_weight=0.0, before_exchange=self.pipe.fire)

    def step(self):
        self.pipe.act_and_step(self._act_env)
        self._timed("learn", self._learn)
        self.pipe.join()
        self.t += 1


def cpu_baseline(args, seconds):
    """The oracle (CPU restatement) timed on this host on a BOUNDED sample of the same workload: the same loop
    (actor forward for all envs, env step for all envs with insert, one HIRL learn at B = 128) for as many vector steps
    as fit in about `seconds`."""
    from oracle import hirl_oracle as H
    from tests import _oracle as ox

    n = args.envs
    rng = np.random.default_rng(0)
    actor, critic, bc = init_params(rng)
    es, ea = synthetic_expert(rng)
    o = H.HirlOracle(actor, critic, bc)
    envs, obs = ox.reset_batch(n, 0, 1, seed=0)
    cap = 1 << 18
    ring = np.zeros((cap, 32), np.float32)
    rs = np.zeros(cap, np.int8)
    total = np.zeros(1, np.uint64)
    epi = np.zeros(n, np.uint32)
    steps, t0 = 0, time.perf_counter()
    while True:
        a = o.choose_action(obs, rng.normal(0, 0.1, (n, 4)).astype(np.float32))
        ox.step_batch(envs, a, obs, max_step=1500, auto_reset=1, randomize=1, seed=0, episode_ctr=epi, ring=ring, ring_succ=rs, total=total)
        m = min(int(total[0]), cap)
        idx = rng.integers(0, m, args.batch)
        rows = ring[idx]
        ibc = rng.integers(0, es.shape[0], args.batch)
        o.learn((rows[:, 0:13], rows[:, 13:17], rows[:, 17:30], rows[:, 30], rows[:, 31]), (es[ibc], ea[ibc]),
                rng.normal(0, 0.2, 4).astype(np.float32), 100 if steps == 0 else o.bc_weight, 0.0)
        steps += 1
        dt = time.perf_counter() - t0
        if dt > seconds or steps >= 2000:
            break
    return {"value": round(n * steps / dt, 1), "unit": "env steps/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "sample": f"{steps} vector steps of {n} envs (oracle: scalar C env step on 1 thread + torch-CPU actor forward and "
                      f"HIRL learn on {torch.get_num_threads()} threads), {dt:.1f} s",
            "update_steps_per_s": round(steps / dt, 2)}


def env_sweep(device):
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
        iters = 30
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            env.step(a)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        out.append({"envs_per_launch": n, "us": round(us, 2), "GBps": round(ENV_BYTES_FUSED * n / us / 1e3, 1),
                    "frac": round(ENV_BYTES_FUSED * n / us / 1e3 / HBM_PEAK_GBPS, 4)})
        del env, rep
        torch.cuda.empty_cache()
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL over xGMI.  HX_BENCH_BACKEND=gloo exists only to exercise this code path where all ranks share one GPU.
        backend = os.environ.get("HX_BENCH_BACKEND", "nccl")
        torch.distributed.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
    loop = Loop(args, rank, world, device)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loop.step()
    # per-stage HIP events (recorded on the stream the kernels are launched on = torch's current stream)
    ar_events = []
    if world > 1:  # exchange step: HIP events around every gradient all-reduce of the first steps (on the stream it is enqueued on)
        inner = loop.eng._allreduce

        def timed_allreduce(t):
            if loop.record and loop.pool:
                a, b = loop.pool.pop(), loop.pool.pop()
                a.record()
                inner(t)
                b.record()
                ar_events.append((t.numel() * 4, a, b))
            else:
                inner(t)
        loop.eng._allreduce = timed_allreduce
    nev = min(args.steps, 512)
    loop.pool = [torch.cuda.Event(enable_timing=True) for _ in range(16 * (nev + 1))]
    L = loop.lib.load()
    loop.kpool = [ctypes.c_void_p(L.hx_event_create()) for _ in range(2 * (nev + 1))]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        loop.record = k >= args.steps - nev  # the last steps of the timed region: clocks settled
        loop.step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    if not loop.krec:  # a run too short to contain a sampled step: take the samples right after the timed region instead
        loop.record, loop.separate = True, True
        for _ in range(16):
            loop.act_env_calls = 7
            loop.step()
        torch.cuda.synchronize()
    med = {k: (float(np.median([a.elapsed_time(b) * 1e3 for a, b in v])) if v else None) for k, v in loop.rec.items()}
    act_us, env_us, learn_us = med["act"], med["env"], med["learn"]
    kern = []
    for a, b in loop.krec:  # kernel-only durations of the env-step launches inside the timed region
        us = ctypes.c_float()
        loop.lib.call("hx_event_elapsed_us", a, b, ctypes.byref(us))
        kern.append(us.value)
    env_kernel_us = float(np.mean(kern))  # mean, like the rocprofv3 --stats average it must agree with
    n_total = args.envs * world
    value = n_total * args.steps / dt
    res = {
        "metric": "env steps/sec (whole node) + HIRL update steps/sec at 4096 envs/GPU", "value": round(value, 1),
        "unit": "env steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 5), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"{args.envs} parallel {args.scenario} envs per GPU, SAC fp32, 1 learn(B={args.batch}) per vector step "
                                f"(BASELINE.json configs[2])" if args.agent == "sac" else
                                f"{args.envs} parallel {args.scenario} envs per GPU, HIRL-{args.type} fp32, 1 learn(B={args.batch}) per vector step "
                                f"(BASELINE.json configs[1])"), "envs_per_gpu": args.envs, "batch": args.batch,
                   "actions": args.actions, "act_env": "two launches" if (loop.separate or loop.uniform) else "one launch (hx_actor_act_step / hx_sac_act_step)", "issue_order": "two streams" if loop.pipe.overlap else "serial", "update_path": "staged" if (args.staged or world > 1) and args.agent == "hirl" else "one-call",
                   "parallelism": f"dp{world}: env shards + replicated nets, RCCL all-reduce of the flat gradients"},
        "update_steps_per_s": round(args.steps / dt, 1),
        "stage_us": {"act+env_step(1 kernel)": None if med["act+env"] is None else round(med["act+env"], 2),
                     "act(own launch, sampled steps)": None if act_us is None else round(act_us, 2),
                     "env_step(own launch, sampled steps)": None if env_us is None else round(env_us, 2),
                     "sample+learn(6-11 kernels)": None if learn_us is None else round(learn_us, 2)},
    }
    # roofline of the env-step kernel (the kernel the metric counts): algorithmic bytes / live-measured launch time
    res["roofline"] = {"kernel": "env_step_kernel<INSERT>", "bound": "hbm", "achieved": round(ENV_BYTES_FUSED * args.envs / env_kernel_us / 1e3, 1),
                       "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ENV_BYTES_FUSED * args.envs / env_kernel_us / 1e3 / HBM_PEAK_GBPS, 4),
                       "traffic": None, "bytes_per_launch": ENV_BYTES_FUSED * args.envs, "us_per_launch": round(env_kernel_us, 2),
                       "launches_timed": len(kern),
                       "timing": "the kernel's own begin/end stamps (hipExtLaunchKernelGGL events) on the launch stream, mean over the "
                                 "timed region's last launches; stage_us brackets the launch with events and so includes the dispatch gap",
                       "traffic_note": "not collectable inside this process"}
    # HBM bytes per launch from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass, gfx950 FETCH x2
    # calibration): a separate run of the same kernel at the same size, committed under profiles/
    pmc_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_env_traffic.json")
    if os.path.exists(pmc_file):
        with open(pmc_file) as f:
            pmc = json.load(f).get(str(args.envs))
        if pmc:
            res["roofline"]["traffic"] = pmc["traffic_bytes"]
            res["roofline"]["traffic_note"] = (f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of tools/pmc_env.py at {args.envs} envs per launch "
                                               f"(profiles/r01p_pmc_*.csv): {pmc['fetch_bytes']} B fetched (FETCH_SIZE x 2, calibrated) + {pmc['write_bytes']} B "
                                               f"written = {pmc['ratio']} x the algorithmic bytes")
    res["roofline_update"] = {"kernels": "fwd_l2/bwd_l2/wgrad/adam (one learn)", "bound": "mfma", "unit": "TFLOP/s",
                              "achieved": round(LEARN_FLOP_PER_SAMPLE * args.batch / learn_us / 1e6, 3), "peak": FP32_MATRIX_PEAK_TFLOPS,
                              "frac": round(LEARN_FLOP_PER_SAMPLE * args.batch / learn_us / 1e6 / FP32_MATRIX_PEAK_TFLOPS, 5),
                              "us_per_learn": round(learn_us, 2)}
    res["roofline_act"] = {"kernels": "act_fused_kernel", "bound": "mfma", "unit": "TFLOP/s",
                           "achieved": round(ACTOR_FLOP * args.envs / act_us / 1e6, 3), "peak": FP32_MATRIX_PEAK_TFLOPS,
                           "frac": round(ACTOR_FLOP * args.envs / act_us / 1e6 / FP32_MATRIX_PEAK_TFLOPS, 5), "us": round(act_us, 2)}
    if world > 1:  # the replicas must still be bit-identical after K sharded updates
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
    res["env_stats"] = loop.env.stats_dict()
    if rank == 0:
        if args.sweep:
            res["roofline_env_sweep"] = env_sweep(device)
        if world == 1 and not args.no_cpu_baseline and args.agent == "hirl":
            res["cpu_baseline"] = cpu_baseline(args, args.cpu_seconds)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
