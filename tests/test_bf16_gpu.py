"""bf16 policy inference (BASELINE.json configs[4]: "bf16 actor/critic + fp32 dynamics") through the C ABI
(hx_actor_act_bf16 / hx_actor_act_step_bf16 / hx_pack_w2_bf16).

Tolerances, stated separately from the fp32 path's 1e-5 (SURVEY.md 7 "bf16 config"):
  * against an fp32 evaluation of the SAME rounded operands — both MATRIX PRODUCTS of the policy beyond its 13-wide input layer take bf16 operands
    since round 5: W2 and h1 (256 -> 512), and W3 and h2 = act(LN2(z2)) (512 -> 4: the final layer rides on the bf16 matrix cores straight from the
    accumulators, hirl4ucav_amd/csrc/hx_act.h "[r5]"); layer 1, both LayerNorms, biases, tanh stay fp32: 99 % of the outputs
    within 1e-4 (only the accumulation order of the dot products differs) and all within 2e-3 (an h1 / h2 element whose fp32
    value sits within an ulp of a bf16 rounding boundary may round the other way in the kernel than in torch: one 2^-9 step);
  * against the full-fp32 policy: |da| <= 2e-2 on the tanh outputs, mean |da| <= 2e-3 (8 significant bits on two operands).
Dynamics, masks, rewards stay exactly what the env step computes from the actions it is given: checked bit for bit."""
import numpy as np
import pytest
import torch.nn.functional as F

from oracle import hirl_oracle as H
from tests import _hirl_data as D
from tests import _oracle as ox

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def image_order(w2):
    """bf16 W2 [512][256] in the order the acting kernel reads it (hx_update.h w2_image_index): 1 KB blocks per (column tile of 16,
    k-slab of 32), inside a block lane (g = k group of 8, r = column) x 8 consecutive k"""
    t = w2.reshape(32, 16, 8, 4, 8)          # [column tile][r][slab][g][e]
    return t.permute(0, 2, 3, 1, 4).reshape(-1)  # [column tile][slab][g][r][e]


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


def rounded_operand_policy(p, x, slope=0.0, layer_norm=True):
    """Actor.forward (HIRL.py:126-140) in fp32 with the operands of the 256 -> 512 and of the 512 -> 4 product rounded to bf16 first
    (slope: F.leaky_relu's, HIRL.py:128-137 `negative_slope`; layer_norm False: the reference's layerNorm=False networks, both norms skipped)."""
    p = {k: torch.as_tensor(v) for k, v in p.items()}
    act = (lambda t: F.leaky_relu(t, slope)) if slope else F.relu
    ln1 = (lambda t: F.layer_norm(t, (256,), p["layernorm1.weight"], p["layernorm1.bias"], 1e-5)) if layer_norm else (lambda t: t)
    ln2 = (lambda t: F.layer_norm(t, (512,), p["layernorm2.weight"], p["layernorm2.bias"], 1e-5)) if layer_norm else (lambda t: t)
    h = act(ln1(F.linear(x, p["full1.weight"], p["full1.bias"])))
    hb = h.to(torch.bfloat16).to(torch.float32)
    w2 = p["full2.weight"].to(torch.bfloat16).to(torch.float32)
    z2 = F.linear(hb.double(), w2.double(), p["full2.bias"].double()).float()  # exact products, fp64 sums: the reference value
    h2 = act(ln2(z2))
    h2b = h2.to(torch.bfloat16).to(torch.float32)
    w3 = p["final.weight"].to(torch.bfloat16).to(torch.float32)
    return torch.tanh(F.linear(h2b.double(), w3.double(), p["final.bias"].double()).float())


def close_to_rounded_operands(a, ref, worst=2e-3):
    d = np.abs(a - ref)
    assert (d <= 1e-4).mean() >= 0.99 and d.max() <= worst, ((d <= 1e-4).mean(), d.max())


@pytest.mark.parametrize("n", [1, 16, 1000, 9000])
def test_bf16_policy_against_rounded_operands_and_fp32(mods, n):
    E = mods[0]
    params = D.make_params(D.PARAM_SEED)
    e = E.HirlEngine(batch=128)
    e.load_params(params["actor"], params["critic"], params["bc_actor"])
    rng = np.random.default_rng(n)
    obs = rng.uniform(-1, 1, (n, 13)).astype(np.float32)
    d_obs = torch.from_numpy(obs).cuda()
    a32 = e.act(d_obs).cpu().numpy()
    e.set_act_dtype("bf16")
    a16 = e.act(d_obs).cpu().numpy()
    # the image is exactly bf16(W2), round to nearest even
    w2 = torch.as_tensor(params["actor"]["full2.weight"]).to(torch.bfloat16)
    assert torch.equal(e.w2_bf16.cpu().view(torch.int16), image_order(w2).view(torch.int16))
    ref = rounded_operand_policy(params["actor"], torch.from_numpy(obs)).numpy()
    close_to_rounded_operands(a16, ref)
    d = np.abs(a16 - a32)
    assert d.max() <= 2e-2 and d.mean() <= 2e-3, (d.max(), d.mean())
    assert d.max() > 0  # it IS the bf16 path
    # noise modes ride on top unchanged
    per = rng.normal(0, 0.3, (n, 4)).astype(np.float32)
    an = e.act(d_obs, noise=torch.from_numpy(per).cuda()).cpu().numpy()
    np.testing.assert_array_equal(an, np.clip(a16 + per, -1, 1).astype(np.float32))


@pytest.mark.parametrize("n", [300, 9000])  # the per-tile kernel, the persistent kernel
@pytest.mark.parametrize("slope,layer_norm", [(0.01, True), (0.0, False), (0.01, False)])
def test_bf16_policy_variants_against_rounded_operands(mods, n, slope, layer_norm):
    """The other two shapes of the reference's networks through the bf16 acting kernels: TD3's leaky slope (HIRL.py:128-137) and layerNorm=False
    (HIRL.py:135-138) — the head that runs LayerNorm 2 + the final layer from the accumulators takes both as template / mode-bit branches."""
    E = mods[0]
    params = D.make_params(D.PARAM_SEED)
    e = E.HirlEngine(batch=128, slope=slope, layer_norm=layer_norm)
    e.load_params(params["actor"], params["critic"], params["bc_actor"])
    e.set_act_dtype("bf16")
    rng = np.random.default_rng(n + 7)
    obs = rng.uniform(-1, 1, (n, 13)).astype(np.float32)
    a16 = e.act(torch.from_numpy(obs).cuda()).cpu().numpy()
    ref = rounded_operand_policy(params["actor"], torch.from_numpy(obs), slope, layer_norm).numpy()
    # without the norms h1 / h2 are not unit-scale (elements of several units): an element that rounds the other way moves by a few 2^-9 steps' worth
    close_to_rounded_operands(a16, ref, worst=2e-3 if layer_norm else 1e-2)
    plain = rounded_operand_policy(params["actor"], torch.from_numpy(obs)).numpy()
    assert np.abs(ref - plain).max() > 1e-3  # the variant is a different function: the check above is not vacuous


def test_bf16_image_follows_the_actor_adam_step(mods):
    """hx_adam(which = 1) rewrites the bf16 image of W2 with the parameters it has just updated: after K learn() calls it equals
    bf16(actor W2) bit for bit, and acting uses the NEW weights."""
    E = mods[0]
    params, data = D.make_params(3), D.make_data(4)
    e = E.HirlEngine(batch=128)
    e.load_params(params["actor"], params["critic"], params["bc_actor"])
    e.set_act_dtype("bf16")
    ring = torch.from_numpy(data["replay"]).cuda()
    bc = np.zeros((D.N_EXPERT, 32), np.float32)
    bc[:, 0:13], bc[:, 13:17] = data["expert_s"], data["expert_a"]
    bc = torch.from_numpy(bc).cuda()
    rng = np.random.default_rng(0)
    obs = torch.from_numpy(rng.uniform(-1, 1, (64, 13)).astype(np.float32)).cuda()
    before = e.act(obs).clone()
    for k in range(5):
        idx = torch.from_numpy(rng.integers(0, D.N_REPLAY, 128).astype(np.int32)).cuda()
        ibc = torch.from_numpy(rng.integers(0, D.N_EXPERT, 128).astype(np.int32)).cuda()
        e.assemble(ring, idx, bc_table=bc, idx_bc=ibc)
        e.learn(noise=torch.from_numpy(rng.normal(0, 0.2, 4).astype(np.float32)).cuda(), bc_weight_now=0.5)
    torch.cuda.synchronize()
    w2 = E.unpack(e.actor, E.ACTOR_LAYOUT)["full2.weight"].to(torch.bfloat16)
    assert torch.equal(e.w2_bf16.view(torch.int16), image_order(w2).to(e.w2_bf16.device).view(torch.int16))
    assert not torch.equal(before, e.act(obs))
    sd = {k: v.cpu().numpy() for k, v in E.unpack(e.actor, E.ACTOR_LAYOUT).items()}
    close_to_rounded_operands(e.act(obs).cpu().numpy(), rounded_operand_policy(sd, obs.cpu()).numpy())


def f32_image_order(w2):
    """fp32 W2 [512][256] in the fp32 image's order (hx_update.h w2f_image_index): 1 KB blocks per (column tile of 16, k-chunk of 16),
    inside a block lane (g = k group of 4, r = column) x 4 consecutive k"""
    return w2.reshape(32, 16, 16, 4, 4).permute(0, 2, 3, 1, 4).reshape(-1)  # [tile][r][chunk][g][e] -> [tile][chunk][g][r][e]


@pytest.mark.parametrize("staged", [False, True])
@pytest.mark.parametrize("n", [1, 16, 1000, 4096, 9000])
def test_f32_image_path_is_bit_identical_and_follows_adam(mods, staged, n):
    """The fp32 policy from the re-ordered image of W2 (hx_actor_act_f32i, what HirlEngine.act uses for its own actor: W2 straight into
    registers, no LDS staging) equals hx_actor_act (row-major W2 streamed through LDS) BIT FOR BIT at every size class (16- and 32-row
    workgroups, ragged tails), and the image follows the actor through the one-call and the staged Adam steps."""
    import ctypes

    from hirl4ucav_amd import _lib

    E = mods[0]
    params, data = D.make_params(5), D.make_data(6)
    e = E.HirlEngine(batch=128)
    e.staged = staged
    e.x9_rows = None  # the fp32-MFMA family at every size (from 4,096 rows on the engine's default is the exact-split format: tests/test_x9_gpu.py)
    e.load_params(params["actor"], params["critic"], params["bc_actor"])
    ring = torch.from_numpy(data["replay"]).cuda()
    bc = np.zeros((D.N_EXPERT, 32), np.float32)
    bc[:, 0:13], bc[:, 13:17] = data["expert_s"], data["expert_a"]
    bc = torch.from_numpy(bc).cuda()
    rng = np.random.default_rng(n)
    obs = torch.from_numpy(rng.uniform(-1, 1, (n, 13)).astype(np.float32)).cuda()

    def plain():
        out = torch.empty((n, 4), dtype=torch.float32, device="cuda")
        _lib.call("hx_actor_act", e.actor.data_ptr(), obs.data_ptr(), n, out.data_ptr(), 0, None, 0.0, 0, 0, 0, e.slope, None, _lib.stream_ptr())
        return out

    assert torch.equal(e.w2_f32i, f32_image_order(E.unpack(e.actor, E.ACTOR_LAYOUT)["full2.weight"]).to(e.w2_f32i.device))
    assert torch.equal(e.act(obs), plain())
    for k in range(4):
        idx = torch.from_numpy(rng.integers(0, D.N_REPLAY, 128).astype(np.int32)).cuda()
        ibc = torch.from_numpy(rng.integers(0, D.N_EXPERT, 128).astype(np.int32)).cuda()
        e.assemble(ring, idx, bc_table=bc, idx_bc=ibc)
        e.learn(noise=torch.from_numpy(rng.normal(0, 0.2, 4).astype(np.float32)).cuda(), bc_weight_now=0.5)
    assert e.actor_step == 2
    assert torch.equal(e.w2_f32i, f32_image_order(E.unpack(e.actor, E.ACTOR_LAYOUT)["full2.weight"]).to(e.w2_f32i.device))
    assert torch.equal(e.act(obs), plain())


@pytest.mark.parametrize("n,scenario", [(4096, "straight_line"), (131072, "mixed")])
def test_bf16_act_step_is_act_then_step(mods, n, scenario):
    """hx_actor_act_step_bf16 (ONE launch at every size: the per-tile kernel up to 8,192 envs, the persistent kernel of hx_actp.hip beyond) == hx_actor_act_bf16 followed by hx_env_step, bit for bit:
    actions, every state word, observations, rewards, masks, statistics, the replay rows as a multiset.  The large case is the
    configured size of BASELINE.json configs[4] (131,072 mixed envs: scenario = id mod 3, sorted) — and its dynamics agree
    with the oracle on a sample, from the actions the bf16 policy produced."""
    E, Env, Replay = mods
    params = D.make_params(D.PARAM_SEED)
    scen = np.sort(np.arange(n) % 3).astype(np.int32) if scenario == "mixed" else scenario
    outs = []
    for fused in (True, False):
        e = E.HirlEngine(batch=128)
        e.load_params(params["actor"], params["critic"], params["bc_actor"])
        e.set_act_dtype("bf16")
        rep = Replay(1 << 20, "cuda")
        env = Env(n, scenario=scen, seed=5, max_step=40, auto_reset=True, random_reset=True, replay=rep)
        env.reset()
        acts = torch.zeros((n, 4), device="cuda")
        hist = []
        for t in range(45):
            prev = env.state.clone() if (t == 44 and not fused) else None
            if fused:
                e.act_step(env, sigma=0.1, seed=11, out=acts)
            else:
                e.act(env.obs, sigma=0.1, seed=11, row0=env.env_id0, out=acts)
                env.step(acts)
            hist.append(acts.clone())
        torch.cuda.synchronize()
        k = int(rep.total.item())
        rows = rep.ring[:min(k, 1 << 20)].cpu().numpy().view(np.uint32)
        rows = rows[np.lexsort(rows.T[::-1])]
        outs.append((torch.stack(hist).cpu().numpy().view(np.uint32), env.state.cpu().numpy().view(np.uint32), env.obs.cpu().numpy().view(np.uint32),
                     env.reward.cpu().numpy().view(np.uint32), env.done.cpu().numpy(), env.success.cpu().numpy(), rows,
                     np.asarray(list(env.stats_dict().values()))))
        if not fused:  # the last step of the two-launch run against the oracle on a sample of envs, from the kernel's own actions
            sample = np.linspace(0, n - 1, 2000).astype(np.int64)
            st = np.ascontiguousarray(prev.cpu().numpy().T[sample])
            o_obs = np.zeros((len(sample), 13), np.float32)
            epi = np.zeros(len(sample), np.uint32)
            a = hist[-1].cpu().numpy()[sample]
            ro, do, so = ox.step_batch(st, a, o_obs, max_step=0, auto_reset=0)
            ended = (env.state[36].view(torch.int32).cpu().numpy()[sample] & 0xFFFF) == 0  # auto-reset envs hold the reset state now
            np.testing.assert_array_equal(env.done.cpu().numpy()[sample], do)
            np.testing.assert_array_equal(env.success.cpu().numpy()[sample], so)
            np.testing.assert_array_equal(env.reward.cpu().numpy()[sample].view(np.uint32), ro.view(np.uint32))
            keep = ~ended
            np.testing.assert_array_equal(env.state.cpu().numpy().T[sample][keep].view(np.uint32), st[keep].view(np.uint32))
            del epi
    for x, y in zip(*outs):
        np.testing.assert_array_equal(x, y)
    assert outs[0][7][7] == 45 * n  # env_steps
