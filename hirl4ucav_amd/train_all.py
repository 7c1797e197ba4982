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


def validate(engine, scenario, episodes, max_step, if_random, seed, device, sac=False):
    """validate() of train_all.py:22-102 (train_sac.py:24-68 for SAC) as ONE batch: `episodes` envs stepped with
    chooseActionNoNoise / exploit until done or the step limit; success / fire success counted only for episodes that ended
    with done (train_all.py:59-64)."""
    env = BatchedHarfangEnv(episodes, scenario=scenario, device=device, seed=seed, auto_reset=False, random_reset=if_random, collect_stats=False)
    obs = env.reset()
    total = torch.zeros(episodes, device=device)
    alive = torch.ones(episodes, dtype=torch.bool, device=device)
    for step in range(max_step):
        a = engine.act(obs, explore=False) if sac else engine.act(obs)
        obs, r, d, s = env.step(a)
        total += torch.where(alive, r, torch.zeros_like(r))
        alive &= d == 0
        if step % 64 == 63 and not bool(alive.any()):
            break
    flags = env.state[35].view(torch.int32)
    done = (flags & _lib.F_DONE) != 0
    success = int((done & ((flags & _lib.F_EPISODE_SUCCESS) != 0)).sum())
    fire = int((done & ((flags & _lib.F_FIRE_SUCCESS) != 0)).sum())
    scores = total.cpu().numpy()
    return float(scores.mean()), float(scores.std()), success, fire


def load_expert(config, rng_seed=0):
    """read_data(data_dir) (train_all.py:222-228); without a file (the Drive data is not available, README.md:6,30) a
    synthetic stand-in of the same shape."""
    if config.expert_csv and os.path.exists(config.expert_csv):
        return read_data(config.expert_csv)
    rng = np.random.default_rng(rng_seed)
    es = rng.uniform(-1, 1, (20000, 13))
    es[:, 7:9] = np.where(rng.random((20000, 2)) < 0.5, 1, -1)
    es[:, 12] = rng.uniform(0, 0.2, 20000)
    ea = rng.uniform(-1, 1, (20000, 4))
    ea[:, 3] = np.where(rng.random(20000) < 1e-3, 1, -1)
    return es, ea


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
    high_score, success_rate, arttir = -math.inf, 0.0, 1
    for episode in range(config.episodes):
        for _ in range(max_step):
            eng.sample(table, None, bc_table, seed=seed + 2)
            eng.bc_train_actor()
        print(f"Episode {episode + 1}: bc_loss {eng.losses_host()[2]:.6f}", flush=True)
        if (episode + 1) % config.checkpoint_rate == 0 and (episode + 1) >= config.bc_validate_from:  # train_all.py:259
            mean, std, succ, fire = validate(eng, env_type, 50, max_step, config.random, seed + 12345, device)
            if mean > high_score or succ / 50 >= success_rate or arttir % 5 == 0:
                torch.save({k: v.cpu().clone() for k, v in E.unpack(eng.actor, E.ACTOR_LAYOUT).items()},
                           os.path.join(model_dir, checkpoint_tag(arttir, succ, 50, mean) + "Actor_Harfang_GYM"))
                high_score, success_rate = max(high_score, mean), max(success_rate, succ / 50)
            print(f"Validation {arttir}: avg reward {mean:.2f} (std {std:.2f}) success {succ / 50:.2f} fire success {fire / 50:.2f}", flush=True)
            arttir += 1
    return log_dir


def main(config):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL over xGMI; HX_DIST_BACKEND=gloo exists only to exercise this path where all ranks share one GPU (tests)
        backend = os.environ.get("HX_DIST_BACKEND", "nccl")
        torch.distributed.init_process_group(backend, **({"device_id": device} if backend == "nccl" else {}))
    if config.seed is not None:
        set_seed(config.seed + rank)
    seed = config.seed or 0
    env_type, n = config.env, config.num_envs
    max_step = MAX_STEP[env_type] * (8 if config.render else 1)
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
        if hirl and config.bc_actor and os.path.exists(config.bc_actor):
            eng.bc_actor.copy_(E.pack(torch.load(config.bc_actor, map_location="cpu"), E.ACTOR_LAYOUT, E.ACTOR_SIZE, device))

    log_dir = os.path.join(config.result_dir, env_type, config.agent, config.model_name, time.strftime("%Y_%m_%d_%H_%M"))
    model_dir = os.path.join(log_dir, "model")
    os.makedirs(model_dir, exist_ok=True)
    run = {"episode": 0, "expert_num": batch if (hirl or esac) else 0, "high_score": -math.inf, "success_rate": 0.0, "arttir": 1}
    if config.resume:  # every rank restores its own shard: <resume>/state_rank<r>.pt
        run = CK.load_run(os.path.join(config.resume, f"state_rank{rank}.pt"), eng, env, replay)
    else:
        # RANDOM EXPLORATION: 20 episodes of uniform actions in the reference (train_all.py:266-282) = 20*maxStep transitions
        env.reset()
        for _ in range(math.ceil(20 * max_step / (n * world))):
            env.step(torch.rand((n, 4), device=device) * 2 - 1)
    expert_num, high_score, success_rate, arttir = run["expert_num"], run["high_score"], run["success_rate"], run["arttir"]
    actions = torch.zeros((n, 4), device=device)
    t0, episode0 = time.time(), run["episode"]
    for episode in range(episode0, config.episodes):
        w_now, warm = bc_weight_schedule(config.type, episode, config.bc_weight) if hirl else (0.0, 0.0)
        for step in range(max_step):
            if config.separate_launches:  # chooseAction, then env.step: two launches
                eng.act(env.obs, seed=seed + 1, row0=env.env_id0, out=actions) if sac else eng.act(env.obs, sigma=0.1, seed=seed + 1, row0=env.env_id0, out=actions)
                env.step(actions)
            elif sac:                     # the same in ONE launch (identical values; only the order in which a step's rows enter the
                eng.act_step(env, seed=seed + 1, out=actions)              # replay ring depends on the workgroup schedule)
            else:
                eng.act_step(env, sigma=0.1, seed=seed + 1, out=actions)
            if step == max_step - 1:
                break
            expert_num = expert_num_after(expert_num, step, warm_up_rate)
            for _ in range(config.updates_per_step):
                if sac:  # train_sac.py:270-273 (SAC) / :401-403 (E-SAC: expert rows mixed in while expert_num > 0)
                    eng.sample(replay, expert if esac else None, n_main=batch - expert_num, seed=seed + 2 + rank)
                    eng.learn()
                    continue
                eng.sample(replay, expert, bc_table, n_main=batch - expert_num, seed=seed + 2 + rank)
                eng.learn(bc_weight_now=w_now, bc_warm_up_weight=warm)
                w_now = None  # afterwards learn()'s own returned weight is fed back (train_all.py:361): the stored device value
        if rank == 0:
            c, a, b, r_, f, w = eng.losses_host()
            st = env.stats_dict()
            sps = (episode + 1 - episode0) * max_step * n * world / (time.time() - t0)
            names = ("q1", "q2", "policy", "entropy_loss", "alpha") if sac else ("critic", "actor", "bc", "rl", "bc_weight")
            vals = (c, a, b, r_, w)
            print(f"Episode {episode + 1}: " + " ".join(f"{k} {v:.4f}" for k, v in zip(names, vals)) + f" | episodes {st['episodes']} "
                  f"kills {st['kills']} fire-success {st['fire_success_episodes']} | {sps:,.0f} env steps/s", flush=True)
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
            arttir += 1
        if world > 1 and (episode + 1) % checkpoint_rate == 0:  # sharded run: the replicas must not have drifted apart
            mine = torch.tensor([eng.replica_checksum()], dtype=torch.int64, device=device)
            every = [torch.zeros_like(mine) for _ in range(world)]
            torch.distributed.all_gather(every, mine)
            if any(int(c.item()) != int(mine.item()) for c in every):
                raise RuntimeError(f"rank {rank}: network replicas diverged at episode {episode + 1}")
        if config.snapshot_every and (episode + 1) % config.snapshot_every == 0:  # whole-run state for --resume (one file per rank)
            CK.save_run(os.path.join(log_dir, f"state_rank{rank}.pt"), eng, env, replay,
                        {"episode": episode + 1, "expert_num": expert_num, "high_score": high_score, "success_rate": success_rate, "arttir": arttir})
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
    p.add_argument("--separate_launches", action="store_true",
                   help="chooseAction and env.step as two launches (default: one fused launch). With <= 256 envs this keeps the replay "
                        "insert order, and so the whole run, reproducible bit for bit")
    p.add_argument("--snapshot_every", type=int, default=25, help="episodes between whole-run snapshots (0: never)")
    p.add_argument("--resume", type=str, default=None, help="run directory holding state_rank<r>.pt to continue from")
    return p


if __name__ == "__main__":
    main(parser().parse_args())
