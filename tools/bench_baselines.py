"""CPU baselines of bench.py (BASELINE.md 3 / SURVEY.md 8d): reported beside the GPU number, never a target.

Test infrastructure in the sense of the oracle rule: these legs (and only these, with tests/ and smoke()) import `oracle/`; the product
(hirl4ucav_amd/) never does.  Every leg runs a BOUNDED sample and says what the sample was."""
import json
import os
import socket
import time

import numpy as np

from tools.bench_inputs import init_params, synthetic_expert


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
        jitter = [int(rng.integers(-100, 101)) for _ in range(3)]
        send("RESET_MACHINE_MATRIX", machine_id=ALLY, position=[jitter[0], 3500 + jitter[1], -4000 + jitter[2]], rotation=[0, 0, 0])
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


def cpu_baselines(args, device):
    """The `cpu_baseline` object of the JSON line (rank 0, N = 1 only): the port of the same loop + B0 / B1 / B2, inside --cpu-seconds."""
    model, cores = host_cpu()
    budget = max(args.cpu_seconds, 1.0)
    if args.agent == "hirl":
        out = baseline_port(args, 0.4 * budget)
        out["host"] = f"{model}, {cores} logical cores"
        out["b0_reference_plumbing"] = baseline_reference_plumbing(0.3 * budget, episodes=args.b0_episodes)
        out["b1_batched_cpu"] = baseline_batched_cpu(0.15 * budget)
        out["b2_eager_rocm_learn"] = baseline_eager_rocm_learn(args, 0.15 * budget, device)
        return out
    out = baseline_port_sac(args, 0.7 * budget)
    out["host"] = f"{model}, {cores} logical cores"
    out["b1_batched_cpu"] = baseline_batched_cpu(0.3 * budget)
    return out
