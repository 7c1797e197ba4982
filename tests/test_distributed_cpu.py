"""world_size-2 gloo tests (CPU) of the N > 1 path: env shards are disjoint and reproducible, and the exchange the engine performs —
ONE message per phase: SUM all-reduce of the flat critic gradient, then of the merged actor message [dL_rl | dL_bc | soft count],
Adam with grad_scale = 1/world, the BC weight from the GLOBAL count — reproduces the single-process update of the global batch.
The arithmetic stand-in on CPU is the update oracle (the engine itself is GPU-only: tests/test_bench_gpu.py and
tests/test_sharded_gpu.py run the real thing with two ranks on a GPU box); the collective sequence is the product's."""
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
    """One sharded learn() as HirlEngine.learn sequences it at world > 1 (agents/engine.py): critic gradient -> ONE all-reduce -> Adam
    with grad_scale 1/world; actor phase -> ONE all-reduce of the merged message [dL_rl | dL_bc | soft count] -> w from the GLOBAL
    count -> g = w dL_bc + (1 - w) dL_rl, scaled 1/world.  The arithmetic stand-in on CPU is the update oracle."""
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
    # ---- critic phase on the local shard ---------------------------------------------------------------------------------------
    s, a, ns, r, d = (torch.tensor(x) for x in (rows[:, 0:13], rows[:, 13:17], rows[:, 17:30], rows[:, 30], rows[:, 31]))
    with torch.no_grad():
        na = (H.actor_forward(o.target_actor, ns) + torch.tensor(noise).clamp(-0.5, 0.5)).clamp(-1, 1)
        q1t, q2t = H.critic_forward(o.target_critic, ns, na)
        y = r.reshape(-1, 1) + 0.99 * torch.min(q1t, q2t) * (1 - d).reshape(-1, 1)
    q1, q2 = H.critic_forward(o.critic, s, a)
    loss = torch.nn.functional.mse_loss(q1, y) + torch.nn.functional.mse_loss(q2, y)
    keys = list(o.critic)
    gl = torch.autograd.grad(loss, [o.critic[k] for k in keys])
    flat = torch.cat([g.reshape(-1) for g in gl])
    dist.all_reduce(flat)                 # exchange 1: HirlEngine._allreduce(grad_critic)
    flat *= 1.0 / world                   # hx_adam(..., grad_scale = 1/world)
    off, grads = 0, {}
    for k, g in zip(keys, gl):
        grads[k] = flat[off:off + g.numel()].reshape(g.shape).clone()
        off += g.numel()
    o.opt_critic.step(o.critic, grads)    # the actor phase sees the UPDATED critic (HIRL.py:288 precedes :296)
    # ---- actor phase: unweighted dL_rl, dL_bc and the local soft count in ONE message ---------------------------------------------
    akeys = list(o.actor)
    pi = H.actor_forward(o.actor, s)
    rl_q = H.critic_q1(o.critic, s, pi)
    g_rl = torch.autograd.grad(-rl_q.mean(), [o.actor[k] for k in akeys])
    bs, ba = torch.tensor(data["expert_s"][ibc[lo:hi]]), torch.tensor(data["expert_a"][ibc[lo:hi]])
    g_bc = torch.autograd.grad(torch.nn.functional.mse_loss(H.actor_forward(o.actor, bs), ba) * 10000.0, [o.actor[k] for k in akeys])
    with torch.no_grad():
        cnt = (H.critic_q1(o.critic, s, H.actor_forward(o.bc_actor, s)) > rl_q).sum().float().reshape(1)
    msg = torch.cat([g.reshape(-1) for g in g_rl] + [g.reshape(-1) for g in g_bc] + [cnt])   # hx_hirl_actor_wgrad_split
    dist.all_reduce(msg)                  # exchange 2: the only collective of the actor phase
    na_ = (msg.numel() - 1) // 2
    w = min(float(msg[-1]) / B, 1.0)      # hx_adam_mixed: count / GLOBAL batch (+ warm = 0), clipped (HIRL.py:308)
    g_actor = (w * msg[na_:2 * na_] + (1.0 - w) * msg[:na_]) * (1.0 / world)
    if rank == 0:
        out["flat"], out["w"], out["loss"], out["g_actor"] = flat.numpy().copy(), w, float(loss), g_actor.numpy().copy()
    dist.destroy_process_group()


def test_gradient_and_soft_count_exchange_matches_global_batch():
    world, port = 2, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_shard_grads, args=(world, port, out), nprocs=world, join=True)
        flat, w, g_actor = out["flat"], out["w"], out["g_actor"]
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
    # two exchanges in all: the soft weight and the mixed actor gradient of the merged message equal the global batch's
    assert abs(w - ret[5]) < 1e-6 and 0 <= w <= 1
    ref_a = np.concatenate([o.last_grads["actor"][k].numpy().ravel() for k in o.actor])
    np.testing.assert_allclose(g_actor, ref_a, rtol=2e-4, atol=2e-5 * np.abs(ref_a).max())


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


class _FakeConn:
    """stands in for RcclDirect over gloo: counts close() calls"""

    def __init__(self, uid, world, rank):
        self.uid, self.world, self.rank, self.closed = uid, world, rank, 0

    def close(self):
        self.closed += 1


def _negotiate(rank, world, port, case, out):
    """exchange.negotiate_rccl_direct (what HirlEngine.use_rccl_direct runs) with one step failing on ONE rank: every rank must come back,
    with the same decision, and a communicator built by the healthy ranks must be closed again (rank 0 failing before the id
    broadcast left the other ranks waiting in it)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import datetime

    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=30))  # a hang fails the test, not the suite
    from hirl4ucav_amd.agents.exchange import negotiate_rccl_direct

    step, bad = case
    made = []

    def boom(what):
        raise RuntimeError(f"{what} fails on rank {rank}")

    def available():
        if step == "available" and rank == bad:
            boom("available")

    def make_id():
        if step == "make_id":
            boom("make_id")
        return b"\x07" * 128

    def connect(uid, w, r):
        assert uid == b"\x07" * 128 and w == world and r == rank
        c = _FakeConn(uid, w, r)
        made.append(c)
        if step == "connect" and rank == bad:
            boom("connect")
        return c

    log = []
    conn, why = negotiate_rccl_direct(None, "cpu", available=available, make_id=make_id, connect=connect, log=log.append)
    out[rank] = (conn is not None, why, len(made), sum(c.closed for c in made), len(log))
    dist.barrier()  # the process group is still usable: nobody is stuck in a collective the others left
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [("none", -1), ("make_id", 0), ("available", 0), ("available", 2), ("connect", 1)])
def test_rccl_direct_negotiation_takes_one_decision_and_never_hangs(case):
    world, port = 3, _free_port()
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_negotiate, args=(world, port, case, out), nprocs=world, join=True)
        res = dict(out)
    step, bad = case
    assert len(res) == world
    if step == "none":
        assert all(r == (True, "", 1, 0, 0) for r in res.values())
        return
    assert all(not ok and why and logged == 1 for ok, why, _, _, logged in res.values())  # everyone falls back, everyone says why
    if step == "connect":  # the healthy ranks built a communicator and gave it back
        assert all(res[r][2] == 1 for r in range(world)) and all(res[r][3] == 1 for r in range(world) if r != bad)
    else:                  # nobody entered the collective init
        assert all(res[r][2] == 0 for r in range(world))
    assert f"rank {bad}" in res[bad][1]
