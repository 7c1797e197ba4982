"""The persistent acting kernel (csrc/hx_actp.hip: beyond 8,192 rows one workgroup per CU keeps W2 and loops over its row tiles, env step of
the same rows in the launch's tail) against the per-tile kernel it replaces there (csrc/hx_act.hip, HX_ACT_PERSIST=0) and against
act-then-step in two launches.  chooseAction for all envs + HarfangEnv.step: hirl/agents/HIRL.py:192-198, hirl/train_all.py:343-345;
SacAgent.explore: hirl/agents/SAC/agent.py:183-190.

Everything here is bit-exact: the two kernels run the same per-row arithmetic (same layer-1 MFMA sequence, same k order of the 256 -> 512
product, same LayerNorm / head functions), only the schedule differs."""
import os

import numpy as np
import pytest

from tests import _hirl_data as D
from tests import _oracle as ox

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mods():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from hirl4ucav_amd import _lib
    from hirl4ucav_amd.agents import engine as E
    from hirl4ucav_amd.environments.batched import BatchedHarfangEnv
    from hirl4ucav_amd.utils.buffer import DeviceReplay

    _lib.load()
    return E, BatchedHarfangEnv, DeviceReplay


class per_tile_kernel:
    """HX_ACT_PERSIST=0 for the calls inside: act_fused_kernel at every size (the library reads the variable per call)"""

    def __enter__(self):
        os.environ["HX_ACT_PERSIST"] = "0"

    def __exit__(self, *a):
        os.environ.pop("HX_ACT_PERSIST", None)
        torch.cuda.synchronize()


def engine(E, dtype, slope=0.0):
    params = D.make_params(D.PARAM_SEED)
    e = E.HirlEngine(batch=128, slope=slope)
    e.load_params(params["actor"], params["critic"], params["bc_actor"])
    if dtype == "bf16":
        e.set_act_dtype("bf16")
    elif dtype == "f32x9":
        e.set_act_dtype("f32x9")
    else:
        e.x9_rows = None  # "f32" here means the fp32-MFMA kernels at every size (the engine's default switches to the exact split at 16,384 rows)
    return e


@pytest.mark.parametrize("dtype", ["bf16", "f32", "f32x9"])
@pytest.mark.parametrize("n", [8193, 8192 + 37, 16384, 40000, 65536 + 5, 131072])
def test_persistent_act_equals_the_per_tile_kernel(mods, dtype, n):
    """every noise mode of chooseAction*: none, one shared draw, a draw per row, Philox N(0, sigma^2) keyed by the global row"""
    E = mods[0]
    e = engine(E, dtype)
    rng = np.random.default_rng(n)
    obs = torch.from_numpy(rng.uniform(-1, 1, (n, 13)).astype(np.float32)).cuda()
    per = torch.from_numpy(rng.normal(0, 0.3, (n, 4)).astype(np.float32)).cuda()
    one = torch.from_numpy(rng.normal(0, 0.3, 4).astype(np.float32)).cuda()
    cases = [dict(), dict(noise=one), dict(noise=per), dict(sigma=0.1, seed=9, row0=77)]
    got = []
    for kw in cases:
        e.act_calls = 10
        got.append(e.act(obs, **kw).clone())
    with per_tile_kernel():
        for kw, g in zip(cases, got):
            e.act_calls = 10
            ref = e.act(obs, **kw)
            assert torch.equal(g.view(torch.int32), ref.view(torch.int32)), (dtype, n, kw.keys())
    assert float(got[0].abs().max()) <= 1.0 and float(got[3].std()) > 0.01


@pytest.mark.parametrize("dtype", ["bf16", "f32"])
def test_persistent_act_with_leaky_slope(mods, dtype):
    """the TD3 agent's activation (slope 0.01): the non-ReLU instantiations"""
    E = mods[0]
    e = engine(E, dtype, slope=0.01)
    n = 20000
    obs = torch.from_numpy(np.random.default_rng(1).uniform(-1, 1, (n, 13)).astype(np.float32)).cuda()
    a = e.act(obs).clone()
    with per_tile_kernel():
        assert torch.equal(a, e.act(obs))


@pytest.mark.parametrize("dtype,n,scenario", [("bf16", 8192 + 40, "mixed"), ("bf16", 16384, "mixed"), ("bf16", 65536, "circular"),
                                              ("f32", 8192 + 40, "mixed"), ("f32", 16384, "serpentine"), ("f32", 65536, "circular"),
                                              ("f32x9", 16384, "mixed")])
def test_persistent_act_step_equals_act_then_step(mods, dtype, n, scenario):
    """ONE launch (persistent kernel, env tail on all waves) == hx_actor_act* followed by hx_env_step: actions, every state word,
    observations, rewards, masks, episode counters, statistics, and the replay rows as a multiset (slots are handed out per workgroup).
    max_step 9 crosses the time limit twice: unstored steps and in-place resets; a ragged last tile at 8,232 envs."""
    E, Env, Replay = mods
    scen = np.sort(np.arange(n) % 3).astype(np.int32) if scenario == "mixed" else scenario
    outs = []
    for fused in (True, False):
        e = engine(E, dtype)
        rep = Replay(1 << 21, "cuda")
        env = Env(n, scenario=scen, seed=5, max_step=9, auto_reset=True, random_reset=True, env_id0=300, replay=rep)
        env.reset()
        acts = torch.zeros((n, 4), device="cuda")
        hist = []
        for t in range(21):
            if fused:
                e.act_step(env, sigma=0.3, seed=11, out=acts)
            else:
                e.act(env.obs, sigma=0.3, seed=11, row0=env.env_id0, out=acts)
                env.step(acts)
            hist.append(acts.clone())
        torch.cuda.synchronize()
        k = int(rep.total.item())
        assert k == n * 21 - n * 2 and k <= rep.capacity
        rows = torch.cat([rep.ring[:k], rep.success[:k].to(torch.float32)[:, None]], 1).cpu().numpy().view(np.uint32)
        rows = rows[np.lexsort(rows.T[::-1])]
        outs.append((torch.stack(hist).cpu().numpy().view(np.uint32), env.state.cpu().numpy().view(np.uint32), env.obs.cpu().numpy().view(np.uint32),
                     env.reward.cpu().numpy().view(np.uint32), env.done.cpu().numpy(), env.success.cpu().numpy(), env.episode_ctr.cpu().numpy(), rows,
                     np.asarray(list(env.stats_dict().values()))))
    for x, y, name in zip(*outs, ("actions", "state", "obs", "reward", "done", "success", "episode_ctr", "replay rows", "stats")):
        np.testing.assert_array_equal(x, y, err_msg=name)
    assert outs[0][8][7] == 21 * n  # env_steps


def test_config3_population_65536_circular_hirl_linear(mods):
    """BASELINE.json configs[3] as ONE population: 65,536 circular envs (HarfangEnv_GYM.py:408-474), actions from the live fp32 policy with
    the exploration noise of train_all.py:343 — size-independent properties over 60 steps with the 40-step limit, and the last step of a
    sample of envs against the oracle bit for bit from the kernel's own actions."""
    E, Env, Replay = mods
    n = 65536
    e = engine(E, "f32")
    e.x9_rows = 16384  # the engine's default: the population's fp32 policy runs through the exact 9-term split
    rep = Replay(1 << 22, "cuda")
    env = Env(n, scenario="circular", seed=7, max_step=40, auto_reset=True, random_reset=True, replay=rep)
    env.reset()
    acts = torch.zeros((n, 4), device="cuda")
    for t in range(59):
        e.act_step(env, sigma=0.1, seed=3, out=acts)
    prev_state, prev_obs = env.state.clone(), env.obs.clone()
    e.act_step(env, sigma=0.1, seed=3, out=acts)
    torch.cuda.synchronize()
    st = env.stats_dict()
    assert st["env_steps"] == 60 * n and st["nonfinite_actions"] == 0
    assert st["episodes"] >= n and st["time_limit"] + st["kills"] <= st["episodes"]  # every env has hit the 40-step limit once
    assert int(rep.total.item()) == 60 * n - st["time_limit"]  # the step that reaches max_step is executed but not stored (train_all.py:346-347)
    flags = env.state[35].view(torch.int32)
    assert bool((((flags >> 8) & 3) == 2).all())  # still circular after the in-place resets
    assert bool(torch.isfinite(env.state[:35]).all()) and bool(torch.isfinite(env.obs).all()) and float(acts.abs().max()) <= 1.0
    k = min(int(rep.total.item()), rep.capacity)
    ring = rep.ring[:k]
    assert bool(((ring[:, 31] == 0) | (ring[:, 31] == 1)).all()) and bool((ring[:, 13:17].abs() <= 1).all())
    # the oracle on a sample of envs: the last step from the state before it and the kernel's own actions
    sample = np.linspace(0, n - 1, 3000).astype(np.int64)
    s0 = np.ascontiguousarray(prev_state.cpu().numpy().T[sample])
    o_obs = np.ascontiguousarray(prev_obs.cpu().numpy()[sample])
    a = acts.cpu().numpy()[sample]
    ro, do, so = ox.step_batch(s0, a, o_obs, max_step=0, auto_reset=0)
    np.testing.assert_array_equal(env.done.cpu().numpy()[sample], do)
    np.testing.assert_array_equal(env.success.cpu().numpy()[sample], so)
    np.testing.assert_array_equal(env.reward.cpu().numpy()[sample].view(np.uint32), ro.view(np.uint32))
    ended = (env.state[36].view(torch.int32).cpu().numpy()[sample] & 0xFFFF) == 0  # auto-reset envs hold the reset state now
    keep = ~ended
    assert keep.sum() > 2000
    np.testing.assert_array_equal(env.state.cpu().numpy().T[sample][keep].view(np.uint32), s0[keep].view(np.uint32))
    np.testing.assert_array_equal(env.obs.cpu().numpy()[sample][keep].view(np.uint32), o_obs[keep].view(np.uint32))


@pytest.mark.parametrize("n", [8192 + 37, 16384, 65536 + 5])
def test_persistent_sac_policy_equals_the_per_tile_kernel(mods, n):
    """SacAgent.explore / exploit (SAC/agent.py:183-196) through the streaming persistent kernel (Gaussian head, fp32 image): exploit, explore
    with given eps, explore with Philox draws — and explore + env step in one launch against two launches (BASELINE.json configs[2] is 16,384
    serpentine envs)."""
    E, Env, Replay = mods
    from hirl4ucav_amd.agents import sac_engine as SE
    from tests.test_oracle_sac import sac_params

    p = sac_params()
    e = SE.SacEngine(batch=128)
    e.load_params(p["policy"], p["q1"], p["q2"])
    e.x9_rows = None  # the fp32-MFMA kernels at every size (the engine's default takes the exact split from 16,384 rows on: next test)
    rng = np.random.default_rng(n)
    obs = torch.from_numpy(rng.uniform(-1, 1, (n, 13)).astype(np.float32)).cuda()
    eps = torch.from_numpy(rng.normal(0, 1, (n, 4)).astype(np.float32)).cuda()
    cases = [dict(explore=False), dict(eps=eps), dict(seed=5, row0=9)]
    got = []
    for kw in cases:
        e.act_calls = 3
        got.append(e.act(obs, **kw).clone())
    with per_tile_kernel():
        for kw, g in zip(cases, got):
            e.act_calls = 3
            assert torch.equal(g.view(torch.int32), e.act(obs, **kw).view(torch.int32)), kw.keys()
    outs = []
    for fused in (True, False):
        e.act_calls = 0
        rep = Replay(1 << 21, "cuda")
        env = Env(n, scenario="serpentine", seed=5, max_step=9, auto_reset=True, random_reset=True, env_id0=40, replay=rep)
        env.reset()
        acts = torch.zeros((n, 4), device="cuda")
        for t in range(12):
            if fused:
                e.act_step(env, seed=11, out=acts)
            else:
                e.act(env.obs, seed=11, row0=env.env_id0, out=acts)
                env.step(acts)
        torch.cuda.synchronize()
        k = int(rep.total.item())
        rows = rep.ring[:k].cpu().numpy().view(np.uint32)
        outs.append((acts.cpu().numpy().view(np.uint32), env.state.cpu().numpy().view(np.uint32), env.obs.cpu().numpy().view(np.uint32),
                     env.reward.cpu().numpy().view(np.uint32), rows[np.lexsort(rows.T[::-1])], np.asarray(list(env.stats_dict().values()))))
    for x, y in zip(*outs):
        np.testing.assert_array_equal(x, y)


def test_sac_policy_exact_split_format_at_large_populations(mods):
    """SacEngine's default from 16,384 rows on: the policy's fp32 256 -> 512 product as the exact 9-term bf16 split (hx_sac_act*_x9, the persistent
    kernel's Gaussian-head instantiation).  Against the fp32-MFMA kernel: the same action within 2e-6 (exact products, another summation order;
    the parity bar against the reference is 1e-5); act + env step in one launch == two launches bit for bit; and the hi | mid | lo images
    follow the policy's optimizer step (hx_sac_learn) exactly: hi + mid + lo == W2."""
    E, Env, Replay = mods
    from hirl4ucav_amd.agents import sac_engine as SE
    from tests.test_oracle_sac import sac_params

    p = sac_params()
    n = 16384
    e, f = SE.SacEngine(batch=128), SE.SacEngine(batch=128)
    f.x9_rows = None
    for x in (e, f):
        x.load_params(p["policy"], p["q1"], p["q2"])
    rng = np.random.default_rng(3)
    obs = torch.from_numpy(rng.uniform(-1, 1, (n, 13)).astype(np.float32)).cuda()
    for kw in (dict(explore=False), dict(seed=5, row0=9)):
        e.act_calls = f.act_calls = 3
        d = (e.act(obs, **kw) - f.act(obs, **kw)).abs().max()
        assert e.w2_x9 is not None and float(d) < 2e-6, float(d)
    outs = []
    for fused in (True, False):
        e.act_calls = 0
        rep = Replay(1 << 20, "cuda")
        env = Env(n, scenario="serpentine", seed=5, max_step=9, auto_reset=True, random_reset=True, env_id0=40, replay=rep)
        env.reset()
        acts = torch.zeros((n, 4), device="cuda")
        for t in range(12):
            if fused:
                e.act_step(env, seed=11, out=acts)
            else:
                e.act(env.obs, seed=11, row0=env.env_id0, out=acts)
                env.step(acts)
        torch.cuda.synchronize()
        rows = rep.ring[:int(rep.total.item())].cpu().numpy().view(np.uint32)
        outs.append((acts.cpu().numpy().view(np.uint32), env.state.cpu().numpy().view(np.uint32), env.obs.cpu().numpy().view(np.uint32),
                     rows[np.lexsort(rows.T[::-1])]))
    for x, y in zip(*outs):
        np.testing.assert_array_equal(x, y)
    # the images after optimizer steps
    rep = Replay(4096, "cuda")
    rep.ring.copy_(torch.from_numpy(rng.normal(size=(4096, 32)).astype(np.float32)))
    rep.ring[:, 31] = (rep.ring[:, 31] > 1.0).float()
    rep.total += 4096
    for k in range(3):
        e.sample(rep, None, seed=11, defer=True)
        e.learn()
    torch.cuda.synchronize()
    w2 = SE.unpack_mlp(e.policy, SE.POLICY_BLOCK, 13, 8)["2.weight"].double().cpu().numpy()
    img = e.w2_x9.view(3, -1).float().cpu().numpy().astype(np.float64)
    np.testing.assert_array_equal(np.sort(img.sum(0)), np.sort(w2.reshape(-1)))  # (sorted: the image order is the kernels' own)
