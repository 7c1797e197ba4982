"""fp32 policy inference with the 256 -> 512 product through the EXACT three-way bf16 split of both operands on the bf16 matrix cores (hx_actor_act_x9 /
hx_actor_act_step_x9, HxNets.actor_w2_x9; Actor.forward HIRL.py:126-140, chooseAction HIRL.py:192-212).  Every fp32 operand is split into
three bf16 parts without loss, the partial products are exact in fp32 and are accumulated in fp32 (six of the nine since round 5: lo x lo, lo x mid, mid x lo lie below the fp32 resolution of the sum — hx_act.h HX_X9_TERMS): the arithmetic of the fp32 kernels
up to the order of the sums.  Bars: the SAME 1e-5 / 1e-6 as the fp32 acting kernel against the oracle and the reference's recorded
actions; its error against an fp64 evaluation no larger than 1.25 x the fp32-MFMA kernel's; the three images reproduce every weight
bit for bit and follow every Adam step of the actor; act + env in one launch == act, then env step."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import hirl_oracle as H  # noqa: E402
from tests import _hirl_data as D  # noqa: E402


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from hirl4ucav_amd.agents import engine
    return engine


def make(E, dtype, params=None, **kw):
    params = params or D.make_params(D.PARAM_SEED)
    e = E.HirlEngine(batch=128, **kw)
    e.load_params(params["actor"], params["critic"], params["bc_actor"])
    e.set_act_dtype(dtype)
    if dtype == "f32":
        e.x9_rows = None  # the fp32-MFMA kernels at every size (the engine's default takes the exact split from 16,384 rows on)
    return e


def f64_actor(p, obs, slope=0.0):
    A = {k: np.asarray(v, np.float64) for k, v in p.items()}

    def ln(x, g, b):
        m = x.mean(-1, keepdims=True)
        v = ((x - m) ** 2).mean(-1, keepdims=True)
        return (x - m) / np.sqrt(v + 1e-5) * g + b

    def act(x):
        return np.where(x > 0, x, slope * x)
    h = act(ln(obs.astype(np.float64) @ A["full1.weight"].T + A["full1.bias"], A["layernorm1.weight"], A["layernorm1.bias"]))
    h = act(ln(h @ A["full2.weight"].T + A["full2.bias"], A["layernorm2.weight"], A["layernorm2.bias"]))
    return np.tanh(h @ A["final.weight"].T + A["final.bias"])


def images_exact(e):
    """hi + mid + lo == the fp32 weight, for every element, and hi is the plain bf16 image"""
    img = e.w2_x9.view(3, -1).float().cpu().numpy().astype(np.float64)
    from hirl4ucav_amd.agents.engine import ACTOR_LAYOUT
    off = dict((k, o) for k, o, _ in ACTOR_LAYOUT)["full2.weight"]
    w2 = e.actor[off:off + 512 * 256].cpu().numpy().astype(np.float64)
    # the image order is the bf16 acting kernel's: pack the same weights the plain way and compare hi with it, the sum with the weights
    ref = torch.zeros(512 * 256, dtype=torch.bfloat16, device="cuda")
    from hirl4ucav_amd import _lib
    _lib.call("hx_pack_w2_bf16", e.actor.data_ptr(), 13, ref.data_ptr(), _lib.stream_ptr())
    torch.cuda.synchronize()
    assert np.array_equal(img[0], ref.float().cpu().numpy().astype(np.float64))
    np.testing.assert_array_equal(np.sort(img.sum(0)), np.sort(w2))  # (sorted: the image order is the kernels' own)


def test_x9_act_matches_oracle_reference_and_fp64(E, golden_dir):
    params = D.make_params(D.PARAM_SEED)
    e, f = make(E, "f32x9", params), make(E, "f32", params)
    images_exact(e)
    o = H.HirlOracle(params["actor"], params["critic"], params["bc_actor"])
    rng = np.random.default_rng(0)
    for n in (1, 5, 16, 1000, 8192 + 40):  # (16-row workgroups at every size)
        obs = rng.uniform(-1, 1, (n, 13)).astype(np.float32)
        a = e.act(torch.from_numpy(obs).cuda()).cpu().numpy()
        np.testing.assert_allclose(a, o.choose_action(obs), rtol=1e-5, atol=1e-6)
        per = rng.normal(0, 0.5, (n, 4)).astype(np.float32)
        a = e.act(torch.from_numpy(obs).cuda(), noise=torch.from_numpy(per).cuda()).cpu().numpy()
        np.testing.assert_allclose(a, o.choose_action(obs, per), rtol=1e-5, atol=1e-6)
    g = np.load(os.path.join(golden_dir, "hirl_choose_action.npz"))
    st = torch.from_numpy(g["states"]).cuda()
    np.testing.assert_allclose(e.act(st).cpu().numpy(), g["action_clean"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(e.act(st, noise=torch.from_numpy(g["noise"]).cuda()).cpu().numpy(), g["action"], rtol=1e-5, atol=1e-6)
    # against fp64: the exact-split kernel is no further away than the fp32-MFMA kernel (it computes the same products, exactly)
    obs = rng.uniform(-1, 1, (16384, 13)).astype(np.float32)
    ref = f64_actor(params["actor"], obs)
    ex = np.abs(e.act(torch.from_numpy(obs).cuda()).cpu().numpy() - ref)
    ef = np.abs(f.act(torch.from_numpy(obs).cuda()).cpu().numpy() - ref)
    assert ex.max() < 1e-6 and ex.mean() <= 1.25 * ef.mean() and ex.max() <= 1.5 * ef.max(), (ex.max(), ex.mean(), ef.max(), ef.mean())
    # the same Philox exploration noise as the fp32 kernel (row, call, seed): the draws do not depend on the arithmetic of the product
    x = torch.from_numpy(obs[:4096]).cuda()
    e.act_calls = f.act_calls = 7
    d = (e.act(x, sigma=0.1, seed=3) - f.act(x, sigma=0.1, seed=3)).abs().max()
    assert float(d) < 2e-6


def test_x9_images_follow_the_actors_adam_steps(E):
    """10 learn() calls (5 actor steps) in the one-call and in the staged path, then BC pre-training steps: the three images still reproduce
    every weight exactly, and the fp32 update itself is bit-identical to an engine without the images (they are written, never read, by it)."""
    for staged in (False, True):
        params = D.make_params(D.PARAM_SEED)
        e, f = make(E, "f32x9", params), make(E, "f32", params)
        e.staged = f.staged = staged
        rng = np.random.default_rng(4)
        for k in range(10):
            rows = torch.from_numpy(D._rows(rng, 128)).cuda()
            bc = torch.from_numpy(D._rows(rng, 128)).cuda()
            noise = torch.from_numpy((0.2 * rng.standard_normal(4)).astype(np.float32)).cuda()
            for eng in (e, f):
                eng.rows.copy_(rows.reshape(-1)); eng.bc_rows.copy_(bc.reshape(-1))
                eng.learn(noise=noise, bc_weight_now=100 if k % 2 else 0.3)
        torch.cuda.synchronize()
        assert torch.equal(e.actor, f.actor) and torch.equal(e.critic, f.critic) and torch.equal(e.target_actor, f.target_actor)
        images_exact(e)
        for eng in (e, f):
            eng.bc_train_actor()
        torch.cuda.synchronize()
        assert torch.equal(e.actor, f.actor)
        images_exact(e)


@pytest.mark.parametrize("n", [83, 4096, 8192 + 40])
def test_x9_act_step_in_one_launch_equals_act_then_step(E, n):
    from hirl4ucav_amd.environments.batched import BatchedHarfangEnv
    from hirl4ucav_amd.utils.buffer import DeviceReplay
    engs, envs, reps = [], [], []
    for _ in range(2):
        engs.append(make(E, "f32x9"))
        rep = DeviceReplay(1 << 19)
        env = BatchedHarfangEnv(n, scenario=np.arange(n) % 3, seed=3, max_step=9, auto_reset=True, random_reset=True, env_id0=100, replay=rep)
        env.reset()
        envs.append(env); reps.append(rep)
    acts = torch.zeros((n, 4), device="cuda")
    for k in range(12):
        engs[0].act(envs[0].obs, sigma=0.3, seed=5, row0=100, out=acts)
        envs[0].step(acts)
        a2 = engs[1].act_step(envs[1], sigma=0.3, seed=5)[0]
        assert torch.equal(acts, a2), f"actions, step {k}"
        for name in ("state", "obs", "reward", "done", "success", "episode_ctr"):
            assert torch.equal(getattr(envs[0], name).view(torch.uint8), getattr(envs[1], name).view(torch.uint8)), f"{name}, step {k}"
        assert torch.equal(envs[0].stats.sum(0), envs[1].stats.sum(0))
    assert int(reps[0].total.item()) == int(reps[1].total.item())


def test_x9_is_an_acting_format_of_its_own(E):
    e = make(E, "f32x9")
    with pytest.raises(ValueError):
        e.set_update_dtype("bf16")
    e.set_act_dtype("f32")
    assert e.nets.actor_w2_x9 is None
    e.set_update_dtype("bf16")
    with pytest.raises(ValueError):
        e.set_act_dtype("f32x9")
