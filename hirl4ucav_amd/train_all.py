#!/usr/bin/env python3
"""Vectorised counterpart of the reference training driver (hirl/train_all.py, HIRL and TD3 branches, :261-487):
the same schedule — random exploration, expert-buffer labelling, per-step act / step / store / learn, expert_num decay,
linear / fixed / soft BC-weight schedules, periodic validation and checkpoints — with the single socket env replaced
by `--num_envs` GPU-resident envs per process and every per-step call replaced by a HIP launch.

    python -m hirl4ucav_amd.train_all --agent HIRL --type soft --env straight_line --random --num_envs 4096
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m hirl4ucav_amd.train_all ...   (one process per GPU)

What "one step" means here: the reference does one learn() per single env step (train_all.py:360-361).  With N envs a
vector step yields N transitions; this driver keeps ONE learn() of the reference's batch size per vector step by
default (`--updates_per_step` raises it).  That ratio is a stated design parameter, not something the reference fixes.
"""
import argparse
import math
import os
import time

import numpy as np
import torch

from . import _lib
from .agents import engine as E
from .agents.HIRL import init_actor_state_dict, init_critic_state_dict
from .environments.batched import SCENARIOS, BatchedHarfangEnv
from .utils import checkpoint as CK
from .utils.buffer import DeviceReplay
from .utils.data_processor import read_data
from .utils.scalars import make_writer
from .utils.seed import set_seed

MAX_STEP = {"straight_line": 1500, "serpentine": 1500, "circular": 1900}  # train_all.py:159-183


# ---- pure host logic (unit-tested on CPU) ----------------------------------------------------------------------------
def bc_weight_schedule(hirl_type, episode, bc_weight, bc_warm_up=False):
    """(bc_weight_now, bc_warm_up_weight) at the start of `episode` — train_all.py:328-339."""
    warm = 0.0
    if hirl_type == "linear":
        return max(bc_weight - episode / 5000, 0), warm
    if hirl_type == "fixed":
        return bc_weight, warm
    if hirl_type == "soft":
        if bc_warm_up:
            warm = max(0.3 - episode / 1000, 0)
        return 100, warm
    raise ValueError(hirl_type)


def expert_num_after(expert_num, step, warm_up_rate=10):
    """train_all.py:356-357: one expert row fewer whenever step % warm_up_rate == 0, down to 0."""
    return expert_num - 1 if (step % warm_up_rate == 0 and expert_num != 0) else expert_num


def expert_pair_indices(done):
    """Which (i, i+1) pairs the reference turns into expert-buffer transitions (train_all.py:289-306): it walks i over the
    expert rows and skips one extra row after a terminal pair.  done[i] = get_termination(states[i+1])."""
    out, i, n = [], 0, len(done) + 1
    while i + 1 < n:
        out.append(i)
        if done[i]:
            i += 1
        i += 1
    return np.asarray(out, np.int64)


def checkpoint_tag(arttir, success, episodes, mean_score):
    return "Agent{}_{}_{}_".format(arttir, round(success / episodes * 100), round(mean_score))  # train_all.py:69


# ---- device side -----------------------------------------------------------------------------------------------------
def label_expert(states, actions, device):
    """Expert replay rows [k, 32] + step_success from consecutive expert states, labelled on the GPU
    (HarfangEnv.get_reward / get_termination, HarfangEnv_GYM.py:299-336)."""
    s = torch.as_tensor(np.asarray(states[:-1], np.float32), device=device).contiguous()
    a = torch.as_tensor(np.asarray(actions[:-1], np.float32), device=device).contiguous()
    ns = torch.as_tensor(np.asarray(states[1:], np.float32), device=device).contiguous()
    m = s.shape[0]
    r = torch.zeros(m, device=device)
    sc = torch.zeros(m, dtype=torch.int8, device=device)
    dn = torch.zeros(m, dtype=torch.uint8, device=device)
    _lib.call("hx_label_transitions", s.data_ptr(), a.data_ptr(), ns.data_ptr(), m, r.data_ptr(), sc.data_ptr(), dn.data_ptr(), _lib.stream_ptr())
    keep = torch.as_tensor(expert_pair_indices(dn.cpu().numpy().astype(bool)), device=device)
    rows = torch.cat([s, a, ns, r[:, None], dn.float()[:, None]], 1)[keep]
    return rows, sc[keep]


def validate(engine, scenario, episodes, max_step, if_random, seed, device, sac=False, env=None):
    """validate() of train_all.py:22-102 (train_sac.py:24-68 for SAC) as ONE batch: `episodes` envs stepped with
    chooseActionNoNoise / exploit until done or the step limit.  The kernels keep simulating an env after its `done` (no auto
    reset here), so everything the reference reads when it SEES done — the score so far, env.episode_success, env.fire_success
    (train_all.py:59-64) — is latched at the first step with done != 0: a missile that is still in flight when the aircraft leaves
    the altitude band must not turn that episode into a kill afterwards.  `env`: a prepared BatchedHarfangEnv (tests)."""
    if env is None:
        env = BatchedHarfangEnv(episodes, scenario=scenario, device=device, seed=seed, auto_reset=False, random_reset=if_random, collect_stats=False)
        obs = env.reset()
    else:
        obs = env.obs
    calls = getattr(engine, "act_calls", None)  # validation must not move the training run's exploration-noise counter
    total = torch.zeros(episodes, device=device)
    alive = torch.ones(episodes, dtype=torch.bool, device=device)
    kill = torch.zeros(episodes, dtype=torch.bool, device=device)
    fired = torch.zeros(episodes, dtype=torch.bool, device=device)
    for step in range(max_step):
        a = engine.act(obs, explore=False) if sac else engine.act(obs)
        obs, r, d, s = env.step(a)
        total += torch.where(alive, r, torch.zeros_like(r))
        ending = alive & (d != 0)
        flags = env.state[35].view(torch.int32)
        kill |= ending & ((flags & _lib.F_EPISODE_SUCCESS) != 0)
        fired |= ending & ((flags & _lib.F_FIRE_SUCCESS) != 0)
        alive &= d == 0
        if step % 64 == 63 and not bool(alive.any()):
            break
    if calls is not None:
        engine.act_calls = calls
    scores = total.cpu().numpy()
    return float(scores.mean()), float(scores.std()), int(kill.sum()), int(fired.sum())


def load_expert(config, rng_seed=0):
    """read_data(data_dir) (train_all.py:222-228).  The reference raises when the file is missing, and so does this: a mistyped path must
    not turn into a run that trains on noise.  `--synthetic_expert` (benchmarks, tests; the Drive data is not available, README.md:6,30)
    asks for a uniform-random stand-in of the same shape explicitly, and says so loudly."""
    if config.expert_csv:
        if not os.path.exists(config.expert_csv):
            raise FileNotFoundError(f"--expert_csv {config.expert_csv}: no such file")
        return read_data(config.expert_csv)
    if not getattr(config, "synthetic_expert", False):
        raise ValueError("HIRL / E-SAC / BC need expert data: pass --expert_csv FILE (hirl4ucav_amd.data.ai_data_col writes one), or "
                         "--synthetic_expert to run on a uniform-random stand-in (throughput runs and tests only)")
    print("WARNING: --synthetic_expert: the expert table is UNIFORM RANDOM noise - throughput / plumbing runs only", flush=True)
    rng = np.random.default_rng(rng_seed)
    es = rng.uniform(-1, 1, (20000, 13))
    es[:, 7:9] = np.where(rng.random((20000, 2)) < 0.5, 1, -1)
    es[:, 12] = rng.uniform(0, 0.2, 20000)
    ea = rng.uniform(-1, 1, (20000, 4))
    ea[:, 3] = np.where(rng.random(20000) < 1e-3, 1, -1)
    return es, ea


def shared_value(value, world):
    """rank 0's `value` on every rank (one broadcast_object_list)"""
    if world <= 1:
        return value
    box = [value]
    torch.distributed.broadcast_object_list(box, src=0)
    return box[0]


def save_hparams(log_dir, **kw):
    """save_parameters_to_txt (train_all.py:104-109,233): log1.txt, one key=value per line"""
    with open(os.path.join(log_dir, "log1.txt"), "w") as f:
        for k, v in kw.items():
            f.write(f"{k}={v}\n")


def log_validation(writer, episode, mean, std, success_rate, fire_rate):
    """the four Validation/* scalars of train_all.py:97-100"""
    writer.add_scalar("Validation/Avg Reward", mean, episode)
    writer.add_scalar("Validation/Std Reward", std, episode)
    writer.add_scalar("Validation/Success Rate", success_rate, episode)
    writer.add_scalar("Validation/Fire Success Rate", fire_rate, episode)
    if hasattr(writer, "flush"):
        writer.flush()


def train_bc(config, device, seed, max_step):
    """The BC branch of train_all.py:244-260: maxStep train_actor() calls per 'episode' on the expert (s, a) table, validation
    every checkpoint_rate episodes from --bc_validate_from on, actor checkpoints with the reference's tag and file name."""
    env_type, batch = config.env, 128
    eng = E.HirlEngine(batch=batch, lr_actor=1e-3, slope=0.01, use_bc=True, device=device)  # BC.py:129-135 leaky_relu
    eng.load_params(init_actor_state_dict(), init_critic_state_dict())
    es, ea = load_expert(config)
    tab = np.zeros((es.shape[0], 32), np.float32)
    tab[:, 0:13], tab[:, 13:17] = es, ea
    bc_table = torch.from_numpy(tab).to(device)
    table = DeviceReplay(es.shape[0], device)  # the sampler wants a main ring as well; the BC step reads only the BC rows
    table.store_rows(bc_table)
    log_dir = os.path.join(config.result_dir, env_type, config.agent, config.model_name, time.strftime("%Y_%m_%d_%H_%M"))
    model_dir = os.path.join(log_dir, "model")
    os.makedirs(model_dir, exist_ok=True)
    save_hparams(log_dir, actorLR=1e-3, batchSize=batch, maxStep=max_step, hiddenLayer1=256, hiddenLayer2=512, agent="BC", model_dir=model_dir,
                 data_dir=config.expert_csv)
    writer = make_writer(os.path.join(log_dir, "summary"))
    high_score, success_rate, arttir = -math.inf, 0.0, 1
    for episode in range(config.episodes):
        for _ in range(max_step):
            eng.sample(table, None, bc_table, seed=seed + 2)
            eng.bc_train_actor()
        bc_loss = eng.losses_host()[2]
        writer.add_scalar("Loss/BC_Loss", bc_loss, (episode + 1) * max_step)  # train_all.py:250-251
        print(f"Episode {episode + 1}: bc_loss {bc_loss:.6f}", flush=True)
        if (episode + 1) % config.checkpoint_rate == 0 and (episode + 1) >= config.bc_validate_from:  # train_all.py:259
            mean, std, succ, fire = validate(eng, env_type, 50, max_step, config.random, seed + 12345, device)
            if mean > high_score or succ / 50 >= success_rate or arttir % 5 == 0:
                torch.save({k: v.cpu().clone() for k, v in E.unpack(eng.actor, E.ACTOR_LAYOUT).items()},
                           os.path.join(model_dir, checkpoint_tag(arttir, succ, 50, mean) + "Actor_Harfang_GYM"))
                high_score, success_rate = max(high_score, mean), max(success_rate, succ / 50)
            print(f"Validation {arttir}: avg reward {mean:.2f} (std {std:.2f}) success {succ / 50:.2f} fire success {fire / 50:.2f}", flush=True)
            log_validation(writer, episode, mean, std, succ / 50, fire / 50)
            arttir += 1
    writer.close()
    return log_dir


def launch_ranks(n, argv):
    """`--gpus N` without a launcher environment: start the N ranks as a CHILD torch.distributed.run (this process has not touched a
    GPU and only relays the exit code); refuse when fewer than N GPUs are visible."""
    import socket
    import subprocess
    import sys

    have = torch.cuda.device_count()  # counting does not initialise the GPU
    if have < n and os.environ.get("HX_DIST_BACKEND", "nccl") == "nccl":
        sys.stderr.write(f"train_all: {n} GPUs requested, {have} visible\n")
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
                            "--master-port", str(port), "-m", "hirl4ucav_amd.train_all"] + list(argv), env=env)


def main(config):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if getattr(config, "gpus", None) and config.gpus != world:
        raise SystemExit(f"train_all: --gpus {config.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not torch.cuda.is_available():
        raise SystemExit("train_all needs a GPU: the product has no CPU path")
    if world > torch.cuda.device_count() and os.environ.get("HX_DIST_BACKEND", "nccl") == "nccl":
        raise SystemExit(f"train_all: {world} ranks but {torch.cuda.device_count()} GPUs visible")
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL over xGMI; HX_DIST_BACKEND=gloo exists only to exercise this path where all ranks share one GPU (tests)
        backend = os.environ.get("HX_DIST_BACKEND", "nccl")
        torch.distributed.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
    # no --seed: rank 0 draws one and every rank uses it (the replicas must start from identical networks either way); it is printed
    # and written to log1.txt, so that "unseeded" differs from --seed 0 and is still reproducible after the fact
    # --resume: the run's seed comes from the snapshot (the Philox keys of acting noise, sampling and random_reset must not change mid-run);
    # an explicit --seed that differs is an error
    snap = CK.read_run(os.path.join(config.resume, f"state_rank{rank}.pt")) if config.resume else None
    snap_seed = snap["driver"].get("seed") if snap is not None else None
    if snap_seed is not None and config.seed is not None and int(config.seed) != int(snap_seed):
        raise SystemExit(f"train_all: --resume {config.resume} was run with seed {snap_seed}, --seed {config.seed} given")
    seed = shared_value(snap_seed if snap_seed is not None else
                        (config.seed if config.seed is not None else int.from_bytes(os.urandom(4), "little") & 0x7FFFFFFF), world)
    set_seed(seed + rank)
    if rank == 0:
        print(f"seed {seed}" + (" (from the snapshot)" if snap_seed is not None else "" if config.seed is not None else " (drawn: no --seed given)"), flush=True)
    env_type, n = config.env, config.num_envs
    max_step = (int(config.max_step) if getattr(config, "max_step", None) else MAX_STEP[env_type]) * (8 if config.render else 1)
    if config.agent == "BC":
        return train_bc(config, device, seed, max_step)
    hirl = config.agent == "HIRL"
    esac = config.agent == "SAC" and config.type == "ESAC"
    batch, buffer_size, checkpoint_rate = 128, config.buffer_size, config.checkpoint_rate  # train_all.py:190-208
    warm_up_rate = 20 if config.agent == "SAC" else 10  # train_sac.py:203 / train_all.py:207

    replay = DeviceReplay(buffer_size, device)
    env = BatchedHarfangEnv(n, scenario=env_type, device=device, seed=seed, max_step=max_step, auto_reset=True,
                            random_reset=config.random, env_id0=rank * n, replay=replay)
    sac = config.agent == "SAC"
    torch.manual_seed(seed)  # identical initial networks on every rank
    if sac:  # train_sac.py:187-215: lr 1e-3, hidden [256, 512], batch 128, target_update_interval 3
        from .agents import sac_engine as SE
        from .agents.SAC.agent import _xavier_mlp

        eng = SE.SacEngine(batch=batch, lr=1e-3, device=device)
        eng.load_params(_xavier_mlp(13, 8), _xavier_mlp(17, 1), _xavier_mlp(17, 1))
    else:
        eng = E.HirlEngine(batch=batch, slope=0.0 if hirl else 0.01, use_bc=hirl, device=device)
        eng.load_params(init_actor_state_dict(), init_critic_state_dict(), init_actor_state_dict() if hirl else None)
    expert_len = bc_len = 0
    expert = bc_table = None
    if hirl or esac:
        es, ea = load_expert(config)
        rows, succ = label_expert(es, ea, device)
        expert = DeviceReplay(rows.shape[0] + 10, device)
        expert.store_rows(rows, succ)
        expert_len, bc_len = rows.shape[0], es.shape[0]
        tab = np.zeros((bc_len, 32), np.float32)
        tab[:, 0:13], tab[:, 13:17] = es, ea
        bc_table = torch.from_numpy(tab).to(device)
        if hirl and config.bc_actor:  # agent.load_bc_actor (train_all.py:311-312): a missing file is an error, as in the reference
            if not os.path.exists(config.bc_actor):
                raise FileNotFoundError(f"--bc_actor {config.bc_actor}: no such file")
            eng.bc_actor.copy_(E.pack(torch.load(config.bc_actor, map_location="cpu"), E.ACTOR_LAYOUT, E.ACTOR_SIZE, device))
        elif hirl and config.type == "soft" and rank == 0:
            print("WARNING: HIRL-soft without --bc_actor: the soft weight is estimated against a randomly initialised bc_actor", flush=True)
    dtype = getattr(config, "dtype", "f32")
    # the run's arithmetic is part of its state: a snapshot taken under one --dtype does not continue under another (the same flag changed
    # meaning once already: round 2's "bf16" was policy inference only, since round 3 it is the policy AND learn()'s 256 <-> 512 products)
    if sac and dtype != "f32":  # (before the snapshot comparison: a SAC run started with --dtype bf16 stored "f32", and resumes with the same command line)
        if rank == 0:
            print(f"WARNING: --dtype {dtype} has no effect on the SAC / E-SAC agents (their kernels are fp32): running fp32", flush=True)
        dtype = "f32"
    snap_dtype = snap["driver"].get("dtype") if snap is not None else None
    if snap_dtype is not None and snap_dtype != dtype:
        raise SystemExit(f"train_all: --resume {config.resume} was run with --dtype {snap_dtype}, --dtype {dtype} given")
    if world > 1 and os.environ.get("HX_DIST_BACKEND", "nccl") == "nccl" and hasattr(eng, "use_rccl_direct"):
        eng.use_rccl_direct()  # ncclAllReduce enqueued by the library on the engine's stream (hx_rccl_*): no torch.distributed call inside learn()
    if dtype == "f32x9" and not sac:  # fp32, the acting kernel's 256 -> 512 product through the exact three-way bf16 split of both operands (engine.set_act_dtype)
        eng.set_act_dtype("f32x9")
    elif dtype != "f32" and not sac:  # bf16: actor AND critic (BASELINE.json configs[4]); bf16_policy: policy inference only
        eng.set_act_dtype("bf16")
        if dtype == "bf16":
            eng.set_update_dtype("bf16")

    # ONE run directory for all ranks: rank 0 names it (a minute boundary between ranks would scatter the state_rank<r>.pt shards)
    log_dir = shared_value(os.path.join(config.result_dir, env_type, config.agent, config.model_name, time.strftime("%Y_%m_%d_%H_%M")), world)
    model_dir = os.path.join(log_dir, "model")
    os.makedirs(model_dir, exist_ok=True)
    writer = None
    if rank == 0:
        save_hparams(log_dir, bufferSize=buffer_size, criticLR=1e-3, actorLR=1e-3, batchSize=batch, maxStep=max_step, validationStep=max_step,
                     hiddenLayer1=256, hiddenLayer2=512, agent=config.agent, model_dir=model_dir, hirl_type=config.type, data_dir=config.expert_csv,
                     num_envs=n, world=world, seed=seed, dtype=getattr(config, "dtype", "f32"))
        writer = make_writer(os.path.join(log_dir, "summary"))
    if config.load_model and not sac:  # agent.loadCheckpoints(tag, model_dir), train_all.py:239-240 (the reference hard-codes the tag)
        src = config.load_dir or model_dir
        for name, flat, layout, size in (("Critic_", eng.critic, E.CRITIC_LAYOUT, E.CRITIC_SIZE), ("Actor_", eng.actor, E.ACTOR_LAYOUT, E.ACTOR_SIZE),
                                         ("TargetCritic_", eng.target_critic, E.CRITIC_LAYOUT, E.CRITIC_SIZE),
                                         ("TargetActor_", eng.target_actor, E.ACTOR_LAYOUT, E.ACTOR_SIZE)):
            f = os.path.join(src, config.load_tag + name + "Harfang_GYM")
            if not os.path.exists(f):
                raise FileNotFoundError(f"--load_model: {f} not found (--load_dir / --load_tag name the checkpoint)")
            flat.copy_(E.pack(torch.load(f, map_location="cpu"), layout, size, device))
        eng.refresh_bf16()
    run = {"episode": 0, "expert_num": batch if (hirl or esac) else 0, "high_score": -math.inf, "success_rate": 0.0, "arttir": 1}
    if config.resume:  # every rank restores its own shard: <resume>/state_rank<r>.pt
        run = CK.load_run(snap, eng, env, replay)
        del snap
    else:
        # RANDOM EXPLORATION: 20 episodes of uniform actions in the reference (train_all.py:266-282) = 20*maxStep transitions
        env.reset()
        for _ in range(math.ceil(20 * max_step / (n * world))):
            env.step(torch.rand((n, 4), device=device) * 2 - 1)
    expert_num, high_score, success_rate, arttir = run["expert_num"], run["high_score"], run["success_rate"], run["arttir"]
    actions = torch.zeros((n, 4), device=device)
    t0, episode0 = time.time(), run["episode"]
    log_rate = 300  # train_all.py:208
    ret = torch.zeros(n, device=device) if config.log_rewards else None
    last_stats = env.stats_dict()
    # --loop front (default where it applies: HIRL / TD3 in fp32 or bf16 actor + critic, batch <= 256, one update per step): the env
    # step and the first two launches of learn() as ONE launch (HirlEngine.step_learn) — the minibatch is then drawn from the ring as it stood before
    # this step's insert.  --loop reference: the reference's order on every step (act -> env step -> insert -> draw -> learn)
    # ... and only with ONE process per GPU: the in-launch waits of the front launch are argued for a chip of its own (include/hirl4ucav.h hx_hirl_front —
    # the waiting workgroups of processes that share a GPU add up; eight front-loop ranks on one GPU tripped in round 5).  HX_FRONT_SHARED_GPU=1 takes it
    # anyway (soak tests); the status word + fallback below stay armed either way.
    shared_gpu = world > torch.cuda.device_count() and not os.environ.get("HX_FRONT_SHARED_GPU")
    front = (config.loop == "front" and not sac and not config.separate_launches and config.updates_per_step == 1 and batch <= 256
             and getattr(config, "dtype", "f32") in ("f32", "f32x9", "bf16") and not shared_gpu)
    front_sac = (config.loop == "front" and sac and world == 1 and not config.separate_launches and config.updates_per_step == 1 and batch <= 256 and n > 8192)
    if rank == 0:
        which = "front launch (env step + first launches of learn() in one launch; draw before the insert)" if (front or front_sac) else "reference order"
        print(f"vector loop: {which}" + (f" ({world} ranks share a GPU: the front launch is for one process per GPU)" if (shared_gpu and config.loop == "front") else ""),
              flush=True)
    # The front launch's in-launch waits (launch B for launch A's rows, launch C for both) assume that the workgroups of ONE launch start in index order —
    # observed on gfx950, promised by nobody.  A wait that gives up sets a sticky status word: it is read every --status_check_every vector steps (one host
    # sync; every rank takes the same decision), and on a trip every rank goes back to the newest snapshot and continues in the reference's order INSIDE this
    # process (--on_front_trip fallback, the default), or the run exits with code 3 (no snapshot yet, or --on_front_trip exit).
    check_every = max(int(config.status_check_every), 0)
    snap_path = os.path.join(config.resume, f"state_rank{rank}.pt") if config.resume else None  # the newest whole-run snapshot this process can go back to
    steps_since_check = 0

    def front_tripped():
        code = eng.front_status()
        if world > 1:
            t = torch.tensor([code], dtype=torch.int32, device=device if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            code = int(t.item())
        return code

    if front and check_every and config.on_front_trip == "fallback" and config.snapshot_every and snap_path is None:
        # a trip before the first periodic snapshot would leave nothing to go back to: take one of the starting state (every rank its shard)
        if world > 1:
            torch.distributed.barrier()
        snap_path = os.path.join(log_dir, f"state_rank{rank}.pt")
        CK.save_run(snap_path, eng, env, replay, {"episode": episode0, "expert_num": expert_num, "high_score": high_score, "success_rate": success_rate,
                                                  "arttir": arttir, "seed": seed, "dtype": dtype})
        if world > 1:
            torch.distributed.barrier()
    episode = episode0
    while episode < config.episodes:
        w_now, warm = bc_weight_schedule(config.type, episode, config.bc_weight) if hirl else (0.0, 0.0)
        tripped = 0
        for step in range(max_step):
            if front and check_every:
                steps_since_check += 1
                if steps_since_check >= check_every:
                    steps_since_check = 0
                    tripped = front_tripped()
                    if tripped:
                        break
            if front and step < max_step - 1:  # (the last step of an episode has no learn() behind it, train_all.py:346-347)
                expert_num = expert_num_after(expert_num, step, warm_up_rate)
                eng.step_learn(env, expert, bc_table, n_main=batch - expert_num, act_sigma=0.1, act_seed=seed + 1, out=actions, sample_seed=seed + 2 + rank,
                               bc_weight_now=w_now, bc_warm_up_weight=warm)
                w_now = None  # afterwards learn()'s own returned weight is fed back (train_all.py:361): the stored device value
                if ret is not None:
                    ret += env.reward
                if writer is not None and step % log_rate == 0:  # Loss/* every 300 steps, train_all.py:362-367
                    c_, a_, b_, r__, f_, _w = eng.losses_host()
                    for tag, v in (("Loss/Critic_Loss", c_), ("Loss/Actor_Loss", a_), ("Loss/BC_Loss", b_), ("Loss/RL_Loss", r__), ("Loss/BC_Fire_Loss", f_)):
                        writer.add_scalar(tag, v, step + episode * max_step)
                continue
            if front_sac and step < max_step - 1:  # SAC / E-SAC: explore + env step + the first forward launch of learn() in one launch (SacEngine.step_learn)
                expert_num = expert_num_after(expert_num, step, warm_up_rate)
                eng.step_learn(env, expert if esac else None, n_main=batch - expert_num, act_seed=seed + 1, out=actions, sample_seed=seed + 2 + rank)
                if ret is not None:
                    ret += env.reward
                continue
            if config.separate_launches:  # chooseAction, then env.step: two launches
                eng.act(env.obs, seed=seed + 1, row0=env.env_id0, out=actions) if sac else eng.act(env.obs, sigma=0.1, seed=seed + 1, row0=env.env_id0, out=actions)
                env.step(actions)
            elif sac:                     # the same in ONE launch (identical values; only the order in which a step's rows enter the
                eng.act_step(env, seed=seed + 1, out=actions)              # replay ring depends on the workgroup schedule)
            else:
                eng.act_step(env, sigma=0.1, seed=seed + 1, out=actions)
            if ret is not None:
                ret += env.reward
            if step == max_step - 1:
                break
            expert_num = expert_num_after(expert_num, step, warm_up_rate)
            for _ in range(config.updates_per_step):
                if sac:  # train_sac.py:270-273 (SAC) / :401-403 (E-SAC: expert rows mixed in while expert_num > 0)
                    eng.sample(replay, expert if esac else None, n_main=batch - expert_num, seed=seed + 2 + rank, defer=True)
                    eng.learn()
                    continue
                eng.sample(replay, expert, bc_table, n_main=batch - expert_num, seed=seed + 2 + rank, defer=True)  # drawn inside learn()'s first launch
                eng.learn(bc_weight_now=w_now, bc_warm_up_weight=warm)
                w_now = None  # afterwards learn()'s own returned weight is fed back (train_all.py:361): the stored device value
            if writer is not None and step % log_rate == 0 and not sac:  # Loss/* every 300 steps, train_all.py:362-367
                c_, a_, b_, r__, f_, _w = eng.losses_host()
                for tag, v in (("Loss/Critic_Loss", c_), ("Loss/Actor_Loss", a_), ("Loss/BC_Loss", b_), ("Loss/RL_Loss", r__), ("Loss/BC_Fire_Loss", f_)):
                    writer.add_scalar(tag, v, step + episode * max_step)
        if front and not tripped:
            tripped = front_tripped()  # ... and at every episode's end: nothing of a tripped episode reaches a validation, a checkpoint or a snapshot
        if tripped:
            why = (f"front launch: an in-launch wait gave up (status word {tripped}: " + " and ".join(w for b_, w in ((1, "launch B waiting for launch A's rows"),
                   (2, "launch C waiting for launches A / B")) if tripped & b_) + f") in episode {episode + 1} — the minibatches read since the last check are suspect")
            if config.on_front_trip != "fallback" or snap_path is None or not os.path.exists(snap_path):
                if rank == 0:
                    print(why + "; " + ("--on_front_trip exit" if config.on_front_trip != "fallback" else "no snapshot to go back to (--snapshot_every)") +
                          ": stopping.  `--loop reference` runs the same update without in-launch waits", flush=True)
                raise SystemExit(3)
            run = CK.load_run(snap_path, eng, env, replay)  # every rank its own shard; all shards of a snapshot come from the same episode
            eng.front_reset()
            expert_num, high_score, success_rate, arttir = run["expert_num"], run["high_score"], run["success_rate"], run["arttir"]
            episode = episode0 = run["episode"]
            t0, last_stats, front = time.time(), env.stats_dict(), False
            if ret is not None:
                ret.zero_()
            if rank == 0:
                print(why + f": back to the snapshot of episode {episode} ({snap_path}), continuing in the REFERENCE's order (no in-launch waits)", flush=True)
                if dtype == "f32" and eng.front_x9 and (eng.x9_rows is None or n < eng.x9_rows):  # [ADVICE r5] say what else changes with the loop
                    print(f"note: the front launch acted in the exact-split format at every size; in the reference's order {n} envs act on fp32 MFMA "
                          f"(the split applies from {eng.x9_rows} rows on): fp32 results up to summation order from here on", flush=True)
                print(f"note: scalars and checkpoints written for episodes {episode + 1}.. before the trip are written again from here", flush=True)
                if writer is not None:
                    writer.add_scalar("Others/Front_trip_rollback_to_episode", episode, episode)
            continue
        if rank == 0:
            c, a, b, r_, f, w = eng.losses_host()
            st = env.stats_dict()
            sps = (episode + 1 - episode0) * max_step * n * world / (time.time() - t0)
            names = ("q1", "q2", "policy", "entropy_loss", "alpha") if sac else ("critic", "actor", "bc", "rl", "bc_weight")
            vals = (c, a, b, r_, w)
            print(f"Episode {episode + 1}: " + " ".join(f"{k} {v:.4f}" for k, v in zip(names, vals)) + f" | episodes {st['episodes']} "
                  f"kills {st['kills']} fire-success {st['fire_success_episodes']} | {sps:,.0f} env steps/s", flush=True)
            if writer is not None:  # Training/* per episode, train_all.py:383-388: rates over the env episodes that ENDED during this driver episode
                ended = max(st["episodes"] - last_stats["episodes"], 1)
                writer.add_scalar("Training/Last 50 Episode Train success rate", (st["kills"] - last_stats["kills"]) / ended, episode)
                writer.add_scalar("Training/Last 50 Episode Fire success rate", (st["fire_success_episodes"] - last_stats["fire_success_episodes"]) / ended, episode)
                writer.add_scalar("Others/BC_weight", w, episode)
                writer.add_scalar("Others/Nonfinite_actions", st.get("nonfinite_actions", 0), episode)
                if ret is not None:
                    tot = float(ret.mean())
                    writer.add_scalar("Training/Episode Reward", tot, episode)
                    writer.add_scalar("Training/Average Step Reward", tot / max_step, episode)
                    ret.zero_()
                last_stats = st
        if (episode + 1) % checkpoint_rate == 0 and rank == 0:  # VALIDATION, train_all.py:400-402 / train_sac.py:431-433
            mean, std, succ, fire = validate(eng, env_type, 50, max_step, config.random, seed + 12345, device, sac)
            if mean > high_score or succ / 50 >= success_rate or arttir % 5 == 0:
                tag = checkpoint_tag(arttir, succ, 50, mean)
                if sac:  # SacAgent.save_models, SAC/agent.py:440-444
                    eng.save_models(model_dir, tag)
                else:    # Agent.saveCheckpoints, HIRL.py:336-342
                    for name, flat, layout in (("Critic_", eng.critic, E.CRITIC_LAYOUT), ("Actor_", eng.actor, E.ACTOR_LAYOUT),
                                               ("TargetCritic_", eng.target_critic, E.CRITIC_LAYOUT), ("TargetActor_", eng.target_actor, E.ACTOR_LAYOUT)):
                        torch.save({k: v.cpu().clone() for k, v in E.unpack(flat, layout).items()}, os.path.join(model_dir, tag + name + "Harfang_GYM"))
                high_score, success_rate = max(high_score, mean), max(success_rate, succ / 50)
            print(f"Validation {arttir}: avg reward {mean:.2f} (std {std:.2f}) success {succ / 50:.2f} fire success {fire / 50:.2f}", flush=True)
            log_validation(writer, episode, mean, std, succ / 50, fire / 50)
            arttir += 1
        # sharded run: the replicas must not have drifted apart
        check_due = (episode + 1) % checkpoint_rate == 0 or (config.replica_check_every and (episode + 1) % config.replica_check_every == 0)
        if world > 1 and check_due:
            mine = torch.tensor([eng.replica_checksum()], dtype=torch.int64, device=device)
            every = [torch.zeros_like(mine) for _ in range(world)]
            torch.distributed.all_gather(every, mine)
            if any(int(c.item()) != int(mine.item()) for c in every):
                raise RuntimeError(f"rank {rank}: network replicas diverged at episode {episode + 1}")
        if config.snapshot_every and (episode + 1) % config.snapshot_every == 0:  # whole-run state for --resume (one file per rank)
            if world > 1:
                torch.distributed.barrier()  # every shard of a snapshot comes from the same episode ...
            CK.save_run(os.path.join(log_dir, f"state_rank{rank}.pt"), eng, env, replay,
                        {"episode": episode + 1, "expert_num": expert_num, "high_score": high_score, "success_rate": success_rate, "arttir": arttir,
                         "seed": seed, "dtype": dtype})
            if world > 1:
                torch.distributed.barrier()  # ... and nobody runs ahead while a shard is still being written
            snap_path = os.path.join(log_dir, f"state_rank{rank}.pt")
        episode += 1
    if writer is not None:
        writer.close()
    if world > 1:
        torch.distributed.destroy_process_group()
    return log_dir


def parser():
    p = argparse.ArgumentParser()  # the reference's flags, train_all.py:489-513
    p.add_argument("--agent", type=str, default="HIRL", choices=["HIRL", "TD3", "BC", "SAC"])  # SAC: the train_sac.py path (D2)
    p.add_argument("--port", type=int, default=None)
    p.add_argument("--type", type=str, default="soft", choices=["soft", "linear", "fixed", "SAC", "ESAC"])  # SAC / ESAC: train_sac.py:440-458
    p.add_argument("--bc_weight", type=float, default=0.5)
    p.add_argument("--model_name", type=str, default="model")
    p.add_argument("--load_model", action="store_true")
    p.add_argument("--render", action="store_true")
    p.add_argument("--plot", action="store_true")
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--env", type=str, default="straight_line", choices=list(SCENARIOS))
    p.add_argument("--random", action="store_true")
    # additions of the vectorised driver
    p.add_argument("--num_envs", type=int, default=4096)
    p.add_argument("--episodes", type=int, default=6000)
    p.add_argument("--updates_per_step", type=int, default=1)
    p.add_argument("--buffer_size", type=int, default=1 << 20)
    p.add_argument("--expert_csv", type=str, default=None)
    p.add_argument("--bc_actor", type=str, default=None)
    p.add_argument("--result_dir", type=str, default="results")
    p.add_argument("--checkpoint_rate", type=int, default=25, help="episodes between validations (train_all.py:206)")
    p.add_argument("--bc_validate_from", type=int, default=1000, help="BC: first episode with validation (train_all.py:259)")
    p.add_argument("--loop", type=str, default="front", choices=["front", "reference"],
                   help="front (default where it applies AND every rank has a GPU of its own: HIRL / TD3 in fp32 or bf16 actor + critic, SAC / E-SAC beyond 8,192 envs on "
                        "one GPU; batch <= 256, one update per step; ranks that share a GPU run the reference's order): env step + the first "
                        "launches of learn() as one launch; the minibatch is drawn from the ring as it stood before the step's insert, without the slots it may "
                        "overwrite.  reference: act -> env step -> insert -> draw -> learn on every step (the minibatch sees this step's transitions)")
    p.add_argument("--separate_launches", action="store_true",
                   help="chooseAction and env.step as two launches (default: one fused launch). With <= 256 envs this keeps the replay "
                        "insert order, and so the whole run, reproducible bit for bit")
    p.add_argument("--snapshot_every", type=int, default=25, help="episodes between whole-run snapshots (0: never)")
    p.add_argument("--resume", type=str, default=None, help="run directory holding state_rank<r>.pt to continue from")
    p.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node; without a launcher environment the driver starts them itself")
    p.add_argument("--dtype", type=str, default="f32", choices=["f32", "bf16", "bf16_policy", "f32x9"],
                   help="f32 (default): fp32 everywhere — from 4,096 envs per GPU on, the ACTING kernel forms its fp32 256->512 product as the exact "
                        "three-way bf16 split of both operands on the bf16 matrix cores (engine.x9_rows); f32x9: that format at every size; bf16: policy inference AND the "
                        "256<->512 products of learn() on bf16 MFMA (fp32 accumulate, fp32 master weights / Adam / LayerNorm / dynamics); bf16_policy: "
                        "policy inference only.  HIRL / TD3 only: the SAC agents run fp32.  Stored in the snapshot: --resume refuses another value")
    p.add_argument("--synthetic_expert", action="store_true", help="uniform-random stand-in for the expert CSV (throughput runs and tests ONLY)")
    p.add_argument("--load_dir", type=str, default=None, help="--load_model: directory of the checkpoint files (default: this run's model dir)")
    p.add_argument("--load_tag", type=str, default="Agent20_successRate0.64", help="--load_model: checkpoint tag (train_all.py:240 hard-codes this one)")
    p.add_argument("--log_rewards", action="store_true", help="also log Training/Episode Reward (one more small launch per vector step)")
    p.add_argument("--max_step", type=int, default=None, help="steps per episode (default: the scenario's, train_all.py:159-183: 1500 / 1500 / 1900); short rehearsal runs")
    p.add_argument("--status_check_every", type=int, default=256,
                   help="front loop: vector steps between reads of the front launch's status word (one host sync each; 0: only at every episode's end)")
    p.add_argument("--on_front_trip", type=str, default="fallback", choices=["fallback", "exit"],
                   help="a front launch's in-launch wait gave up: fallback = reload the newest snapshot and continue with --loop reference inside this process "
                        "(exit code 3 when there is no snapshot yet); exit = always exit with code 3")
    p.add_argument("--replica_check_every", type=int, default=5, help="sharded runs: episodes between replica checksum comparisons (0: only at validation)")
    return p


if __name__ == "__main__":
    import sys

    cfg = parser().parse_args()
    if cfg.gpus and cfg.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(cfg.gpus, sys.argv[1:]))
    main(cfg)
