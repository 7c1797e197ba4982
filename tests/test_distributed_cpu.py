"""world_size-2 gloo tests (CPU) of the N > 1 path: env shards are disjoint and reproducible, and the gradient /
soft-count exchange the engine performs (SUM all-reduce of the flat gradient, then Adam with grad_scale = 1/world;
SUM all-reduce of the soft count divided by the global batch) reproduces the single-process update of the global batch.
The arithmetic stand-in on CPU is the update oracle; the collective sequence is the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import hirl_oracle as H
from tests import _hirl_data as D
from tests import _oracle as ox


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_grads(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    params, data = D.make_params(3), D.make_data(4)
    B = 128
    rng = np.random.default_rng(0)
    idx, ibc = rng.integers(0, D.N_REPLAY, B), rng.integers(0, D.N_EXPERT, B)
    noise = rng.normal(0, 0.2, 4).astype(np.float32)
    lo, hi = rank * B // world, (rank + 1) * B // world
    o = H.HirlOracle(params["actor"], params["critic"], params["bc_actor"])
    rows = data["replay"][idx[lo:hi]]
    # ---- critic phase on the local shard; then exactly what HirlEngine.learn does with the flat gradient -------------
    s, a, ns, r, d = (torch.tensor(x) for x in (rows[:, 0:13], rows[:, 13:17], rows[:, 17:30], rows[:, 30], rows[:, 31]))
    with torch.no_grad():
        na = (H.actor_forward(o.target_actor, ns) + torch.tensor(noise).clamp(-0.5, 0.5)).clamp(-1, 1)
        q1t, q2t = H.critic_forward(o.target_critic, ns, na)
        y = r.reshape(-1, 1) + 0.99 * torch.min(q1t, q2t) * (1 - d).reshape(-1, 1)
    q1, q2 = H.critic_forward(o.critic, s, a)
    loss = torch.nn.functional.mse_loss(q1, y) + torch.nn.functional.mse_loss(q2, y)
    keys = list(o.critic)
    flat = torch.cat([g.reshape(-1) for g in torch.autograd.grad(loss, [o.critic[k] for k in keys])])
    dist.all_reduce(flat)                 # HirlEngine._allreduce(grad_critic)
    flat *= 1.0 / world                   # hx_adam(..., grad_scale = 1/world)
    # ---- soft count: SUM all-reduce, divided by the GLOBAL batch ------------------------------------------------------
    with torch.no_grad():
        pi = H.actor_forward(o.actor, s)
        cnt = (H.critic_q1(o.critic, s, H.actor_forward(o.bc_actor, s)) > H.critic_q1(o.critic, s, pi)).sum().to(torch.int32).reshape(1)
    dist.all_reduce(cnt)                  # HirlEngine._allreduce(soft_count)
    w = float(cnt.item()) / B             # hx_hirl_actor_wgrad(count_batch = B * world)  [B here is the global batch]
    if rank == 0:
        out["flat"], out["w"], out["loss"] = flat.numpy().copy(), w, float(loss)
    dist.destroy_process_group()


def test_gradient_and_soft_count_exchange_matches_global_batch():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_shard_grads, args=(world, port, out), nprocs=world, join=True)
        flat, w = out["flat"], out["w"]
    # single process, global batch
    params, data = D.make_params(3), D.make_data(4)
    rng = np.random.default_rng(0)
    idx, ibc = rng.integers(0, D.N_REPLAY, 128), rng.integers(0, D.N_EXPERT, 128)
    noise = rng.normal(0, 0.2, 4).astype(np.float32)
    o = H.HirlOracle(params["actor"], params["critic"], params["bc_actor"])
    rows = data["replay"][idx]
    ret = o.learn((rows[:, 0:13], rows[:, 13:17], rows[:, 17:30], rows[:, 30], rows[:, 31]), (data["expert_s"][ibc], data["expert_a"][ibc]), noise, 100, 0.0)
    ref = np.concatenate([o.last_grads["critic"][k].numpy().ravel() for k in o.critic])
    np.testing.assert_allclose(flat, ref, rtol=1e-4, atol=1e-5 * np.abs(ref).max())
    # the oracle computed the soft weight with the critic AFTER its Adam step; recompute the pre-step count for the check
    o2 = H.HirlOracle(params["actor"], params["critic"], params["bc_actor"])
    s = torch.tensor(rows[:, 0:13])
    with torch.no_grad():
        c = (H.critic_q1(o2.critic, s, H.actor_forward(o2.bc_actor, s)) > H.critic_q1(o2.critic, s, H.actor_forward(o2.actor, s))).float().mean().item()
    assert abs(w - c) < 1e-9 and 0 <= ret[5] <= 1


def _shard_envs(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 64
    envs, obs = ox.reset_batch(n, 0, 1, seed=5, env_id0=rank * n)  # the bench / driver give shard r env ids [r n, (r+1) n)
    t = torch.from_numpy(envs[:, 0:3].copy())
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    steps = torch.tensor([n * 10.0])      # per-rank env steps; whole-job value = SUM over ranks / MAX time
    dist.all_reduce(steps)
    tt = torch.tensor([1.0 + rank])
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    if rank == 0:
        out["pos"] = torch.cat(gathered).numpy()
        out["steps"], out["t"] = float(steps), float(tt)
    dist.destroy_process_group()


def test_env_shards_are_disjoint_streams_of_one_global_env_set():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_shard_envs, args=(world, port, out), nprocs=world, join=True)
        pos, steps, t = out["pos"], out["steps"], out["t"]
    whole, _ = ox.reset_batch(128, 0, 1, seed=5, env_id0=0)
    np.testing.assert_array_equal(pos, whole[:, 0:3])  # two shards of 64 == one set of 128: no collective on the data path
    assert steps == 1280.0 and t == 2.0
