"""The bf16 UPDATE path (BASELINE.json configs[4] "bf16 actor/critic + fp32 dynamics"; include/hirl4ucav.h "bf16 update path"):
HxNets.w2_bf16_all set -> every learn() runs z2 = bf16(h1) bf16(W2)^T, dh1 = bf16(dz2) bf16(W2), dW2 = bf16(dz2)^T bf16(h1) of all
five networks on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; master weights, Adam, LayerNorm, layer 1, heads, targets, losses fp32.

Checked through the C ABI against the ROUNDED-OPERAND oracle (oracle/hirl_oracle.py inside `with Bf16Layer2()`: the same three products
with both operands rounded to bf16, exact products, fp64 sums — Agent.learn, HIRL.py:221-334, otherwise unchanged), call by call from
identical states, 10 consecutive calls x {soft, soft + expert rows, fixed, linear}:

  returned losses        rtol 1e-4 (atol 2e-5)
  gradients, per tensor  >= 99.8 % of the entries within |dg| <= 2e-3 |g| + 2e-4 max|g|, EVERY entry within 2e-2 |g| + 2e-3 max|g|
  parameters / targets   every entry within 2e-6 + 2.1e-3 min(1, 8 tol/|g|) of the oracle's step, tol = the loose gradient bound
                         (Adam turns an unresolved gradient into +- lr)
  images                 bit for bit bf16(fp32 master) after every call: forward, transposed and target images

Two things the comparison has to account for (both are properties of rounding, not of the kernels):
  * the kernel and the oracle agree on every ROUNDED operand except where an fp32 value sits within an ulp of a bf16 rounding boundary
    (about one h1 / dz2 element in 2 x 10^4): that element differs by 2^-9 relative and its row's z2 by ~5e-4 — hence 2e-3 / 2e-2
    instead of the fp32 path's 1e-4;
  * a hidden unit whose pre-activation lies within that noise of zero takes one subgradient in the kernel and the other in the oracle,
    which moves whole rows of gradient entries by a visible fraction.  The activation is continuous there, so both are valid; the
    oracle is therefore evaluated with the KERNEL's choice at every such unit (KernelKinks: the kernel's masks are read back from its
    workspace), and the test asserts that every unit where the two disagree has |pre-activation| <= 5e-3 (LayerNorm outputs are O(1)).
The fp32 path's bars (tests/test_hirl_gpu.py: losses 2e-5, gradients 1e-4 |g| + 2e-5 max|g|) are unchanged and its results bit-identical
with the bf16 machinery compiled in (this file's last test)."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import hirl_oracle as H  # noqa: E402
from tests import _hirl_data as D  # noqa: E402
from tests.test_hirl_gpu import device_tables, sync_oracle  # noqa: E402

LOSS_RTOL, LOSS_ATOL = 1e-4, 2e-5
TIGHT = (2e-3, 2e-4)   # |dg| <= a |g| + b max|g| for >= 99.8 % of a tensor's entries
LOOSE = (2e-2, 2e-3)   # ... and for every entry
KINK = 5e-3            # a unit may take the kernel's subgradient only if the oracle's pre-activation is this close to zero

# workspace slot layout (hx_update.h Slot / carve_slot, S_* order)
XP, H1, H2, OW, KC = 20, 256, 512, 8, 8
SLOT_FIELDS = (("x", XP), ("z1", H1), ("st1", 2), ("h1", H1), ("z2", H2), ("st2", 2), ("outv", OW), ("dz2", H2), ("dh1", H1), ("dout", OW), ("lnp", 2 * KC))
SLOTS = ["TA", "C1", "C2", "TC1", "TC2", "API", "ABC", "BCS", "CPI", "CSOFT"]


def slot_fields(e, name, B=128):
    per_row = sum(n for _, n in SLOT_FIELDS)
    base = SLOTS.index(name) * per_row * B
    ws = e.ws[base:base + per_row * B].cpu()
    out, o = {}, 0
    for k, n in SLOT_FIELDS:
        out[k] = ws[o:o + B * n].reshape(B, n)
        o += B * n
    return out


def kernel_masks(e, name, ln2_w, ln2_b):
    """(layer-1 mask, layer-2 mask) the kernel used for the net evaluated in slot `name`: h1 > 0 as saved; y2 = g2 xhat2 + be2 > 0
    from the saved z2 and LayerNorm statistics with the LayerNorm parameters in force at that forward pass"""
    f = slot_fields(e, name)
    xhat = (f["z2"] - f["st2"][:, 0:1]) * f["st2"][:, 1:2]
    return f["h1"] > 0, (ln2_w * xhat + ln2_b) > 0


class KernelKinks:
    """Patches the oracle's activation: call number c of a learn() (the order the oracle evaluates its activations in) takes the
    given mask wherever the oracle's own pre-activation is within KINK of zero and the two masks differ; a disagreement farther from zero
    is an error.  Forward and backward use the same mask (|x| <= KINK there: the forward value moves by at most that)."""

    def __init__(self, masks):
        self.masks, self.calls, self.taken = masks, 0, 0

    def __enter__(self):
        self._orig = H._act

        class Masked(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, slope, mask):
                ctx.slope, ctx.mask = slope, mask
                return torch.where(mask, x, x * slope)

            @staticmethod
            def backward(ctx, g):
                return g * torch.where(ctx.mask, torch.ones_like(g), torch.full_like(g, ctx.slope)), None, None

        def act(x, slope=0.0):
            c = self.calls
            self.calls += 1
            if c not in self.masks:
                return self._orig(x, slope)
            own = (x > 0).detach()
            differ = own != self.masks[c]
            if not differ.any():
                return self._orig(x, slope)
            far = differ & (x.detach().abs() > KINK)
            assert not far.any(), f"activation call {c}: the kernel's mask differs at |pre-activation| {float(x.detach().abs()[far].max()):.3e} > {KINK}"
            self.taken += int(differ.sum())
            return Masked.apply(x, slope, torch.where(differ, self.masks[c], own))

        H._act = act
        return self

    def __exit__(self, *exc):
        H._act = self._orig
MODES = ["soft_e0", "soft_e64", "fixed_e32", "linear_e0"]


@pytest.fixture(scope="module")
def E():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from hirl4ucav_amd.agents import engine

    return engine


def fwd_image(w2):
    """[512][256] -> w2_image_index order (hx_update.h): blocks (column tile, k slab) x lane (g, r) x 8 k"""
    return w2.reshape(32, 16, 8, 4, 8).permute(0, 2, 3, 1, 4).reshape(-1)


def t_image(w2):
    """[512][256] -> w2t_image_index order: blocks (tile of 16 k1, slab of 32 n) x lane (g, r = k1) x 8 n"""
    return w2.t().reshape(16, 16, 16, 4, 8).permute(0, 2, 3, 1, 4).reshape(-1)


def expected_images(e, E_):
    """every image of the block from the engine's fp32 masters, in IM_* order (hx_update.h)"""
    a = E_.unpack(e.actor, E_.ACTOR_LAYOUT)["full2.weight"]
    ta = E_.unpack(e.target_actor, E_.ACTOR_LAYOUT)["full2.weight"]
    bc = E_.unpack(e.bc_actor, E_.ACTOR_LAYOUT)["full2.weight"]
    c, tc = E_.unpack(e.critic, E_.CRITIC_LAYOUT), E_.unpack(e.target_critic, E_.CRITIC_LAYOUT)
    b = lambda w: w.to(torch.bfloat16)  # noqa: E731
    parts = [fwd_image(b(a)), fwd_image(b(c["full2.weight"])), fwd_image(b(c["full4.weight"])), fwd_image(b(ta)),
             fwd_image(b(tc["full2.weight"])), fwd_image(b(tc["full4.weight"])), fwd_image(b(bc)),
             t_image(b(a)), t_image(b(c["full2.weight"])), t_image(b(c["full4.weight"]))]
    return torch.cat(parts)


def assert_images_current(e, E_, what):
    got, exp = e.images.view(torch.int16), expected_images(e, E_).view(torch.int16)
    if not torch.equal(got, exp):
        n = 512 * 256
        names = ["actor", "critic1", "critic2", "target_actor", "target_critic1", "target_critic2", "bc_actor", "actor^T", "critic1^T", "critic2^T"]
        bad = [names[i] for i in range(10) if not torch.equal(got[i * n:(i + 1) * n], exp[i * n:(i + 1) * n])]
        raise AssertionError(f"{what}: stale bf16 images {bad}")


def grad_stats(got_flat, oracle_grads, layout):
    """per tensor: fraction of entries outside TIGHT, worst entry relative to the LOOSE bound"""
    got = got_flat.cpu().numpy()
    out = []
    for k, off, shp in layout:
        g = oracle_grads[k].numpy().ravel()
        x = got[off:off + g.size]
        gm = max(np.abs(g).max(), 1e-30)
        d = np.abs(x - g)
        out.append((k, float((d > TIGHT[0] * np.abs(g) + TIGHT[1] * gm).mean()), float((d / (LOOSE[0] * np.abs(g) + LOOSE[1] * gm)).max())))
    return out


def check_grads(got_flat, oracle_grads, layout, what):
    for k, frac_tight, worst_loose in grad_stats(got_flat, oracle_grads, layout):
        assert frac_tight <= 0.002, f"{what} {k}: {frac_tight:.4f} of the entries outside |dg| <= {TIGHT[0]} |g| + {TIGHT[1]} max|g|"
        assert worst_loose <= 1.0, f"{what} {k}: an entry at {worst_loose:.2f} x the loose bound {LOOSE}"


def check_params(e, o, E_, what, was_actor_call):
    for flat, layout, ref, grads in ((e.actor, E_.ACTOR_LAYOUT, o.actor, o.last_grads.get("actor") if was_actor_call else None),
                                     (e.critic, E_.CRITIC_LAYOUT, o.critic, o.last_grads["critic"]),
                                     (e.target_actor, E_.ACTOR_LAYOUT, o.target_actor, o.last_grads.get("actor") if was_actor_call else None),
                                     (e.target_critic, E_.CRITIC_LAYOUT, o.target_critic, o.last_grads["critic"])):
        f = flat.cpu().numpy()
        for k, off, shp in layout:
            r = ref[k].detach().numpy().ravel()
            d = np.abs(f[off:off + r.size] - r)
            if grads is None:
                bound = np.full_like(d, 2e-6)
            else:
                g = np.abs(grads[k].numpy().ravel())
                tol = LOOSE[0] * g + LOOSE[1] * max(g.max(), 1e-30)
                bound = 2e-6 + 2.1e-3 * np.minimum(1.0, 8.0 * tol / np.maximum(g, 1e-30))
            assert not (d > bound).any(), f"{what} {k}: {int((d > bound).sum())} entries beyond their bound, worst {d.max():.2e}"


def learn_masks(e, E_, pre, was_actor_call, soft, use_bc=True):
    """{activation call number of HirlOracle.learn: the kernel's mask} for the calls whose subgradient reaches a parameter gradient:
    calls 0-5 are the target networks (no gradient), 6/7 8/9 the critic's two heads, then on an actor call 10/11 actor(s), 12/13
    Q1(s, pi) with the UPDATED critic, [14-17 the soft estimate, no gradient], and last actor(s_bc)."""
    pre_actor, pre_critic = (E_.unpack(pre[0], E_.ACTOR_LAYOUT), E_.unpack(pre[1], E_.CRITIC_LAYOUT))
    now_critic = E_.unpack(e.critic, E_.CRITIC_LAYOUT)
    cpu = lambda t: t.cpu()  # noqa: E731
    m = {}
    m[6], m[7] = kernel_masks(e, "C1", cpu(pre_critic["layernorm2.weight"]), cpu(pre_critic["layernorm2.bias"]))
    m[8], m[9] = kernel_masks(e, "C2", cpu(pre_critic["layernorm4.weight"]), cpu(pre_critic["layernorm4.bias"]))
    if was_actor_call:
        m[10], m[11] = kernel_masks(e, "API", cpu(pre_actor["layernorm2.weight"]), cpu(pre_actor["layernorm2.bias"]))
        m[12], m[13] = kernel_masks(e, "CPI", cpu(now_critic["layernorm2.weight"]), cpu(now_critic["layernorm2.bias"]))
        if use_bc:
            c = 18 if soft else 14
            m[c], m[c + 1] = kernel_masks(e, "ABC", cpu(pre_actor["layernorm2.weight"]), cpu(pre_actor["layernorm2.bias"]))
    return m


def run_mode(E_, mode, golden_dir, staged=False, collect=None):
    g = np.load(os.path.join(golden_dir, f"hirl_learn_{mode}.npz"))  # (only the recorded minibatch indices, noise and weights are used)
    params, data = D.make_params(D.PARAM_SEED), D.make_data(D.DATA_SEED)
    ring, exp, bc = device_tables(data)
    e = E_.HirlEngine(batch=128)
    e.staged = staged
    e.load_params(params["actor"], params["critic"], params["bc_actor"])
    e.set_update_dtype("bf16")
    assert_images_current(e, E_, "after set_update_dtype")
    o = H.HirlOracle(params["actor"], params["critic"], params["bc_actor"])
    o.split_actor_grads = True
    ne = int(g["expert_num"])
    all_losses, kinks = [], 0
    for k in range(g["out"].shape[0]):
        idx = np.concatenate([g["idx_buf"][k], g["idx_exp"][k]]).astype(np.int32)
        w_in = 100 if g["bc_w_in"][k] == 100 else float(g["bc_w_in"][k])
        warm = float(g["warm_in"][k])
        was_actor_call = e.actor_trainable
        sync_oracle(o, e, E_)
        pre = (e.actor.clone(), e.critic.clone())  # the parameters the critic-phase forwards and the actor's own forwards run with
        e.assemble(ring, torch.from_numpy(idx).cuda(), expert_ring=exp, n_main=128 - ne, bc_table=bc,
                   idx_bc=torch.from_numpy(g["idx_bc"][k].astype(np.int32)).cuda())
        e.learn(noise=torch.from_numpy(g["noise"][k]).cuda(), bc_weight_now=w_in, bc_warm_up_weight=warm)
        got = e.losses_host()
        all_losses.append(got)
        rows = data["replay"][g["idx_buf"][k]]
        if ne:
            rows = np.concatenate([rows, data["expert_rows"][g["idx_exp"][k]]], 0)
        ob = (rows[:, 0:13], rows[:, 13:17], rows[:, 17:30], rows[:, 30], rows[:, 31])
        with H.Bf16Layer2(), KernelKinks(learn_masks(e, E_, pre, was_actor_call, soft=(w_in == 100))) as kk:
            ref = o.learn(ob, (data["expert_s"][g["idx_bc"][k]], data["expert_a"][g["idx_bc"][k]]), g["noise"][k], w_in, warm)
        kinks += kk.taken
        what = f"bf16 {mode} call {k}"
        if collect is not None:
            collect.append((what, got, ref, grad_stats(e.grad_critic, o.last_grads["critic"], E_.CRITIC_LAYOUT),
                            grad_stats(e.grad_actor, o.last_grads["actor"], E_.ACTOR_LAYOUT) if was_actor_call else None))
        else:
            np.testing.assert_allclose(got, ref, rtol=LOSS_RTOL, atol=LOSS_ATOL, err_msg=what + " losses vs the rounded-operand oracle")
            check_grads(e.grad_critic, o.last_grads["critic"], E_.CRITIC_LAYOUT, what + " critic gradient")
            if was_actor_call:
                check_grads(e.grad_actor, o.last_grads["actor"], E_.ACTOR_LAYOUT, what + " actor gradient")
            check_params(e, o, E_, what + " parameters", was_actor_call)
            # ... and the bf16 result is the fp32 result to bf16 accuracy (the reference's recorded fp32 run)
            np.testing.assert_allclose(got[0], g["out"][k][0], rtol=5e-2, err_msg=what + " critic loss vs the fp32 golden")
        assert_images_current(e, E_, what)
    assert e.critic_step == 10 and e.actor_step == 5 and e.update_count == 5
    if collect is not None:
        collect.append(("units that took the kernel's subgradient", kinks))
    return e, all_losses


@pytest.mark.parametrize("mode", MODES)
def test_bf16_learn_matches_the_rounded_operand_oracle(E, mode, golden_dir):
    run_mode(E, mode, golden_dir)


@pytest.mark.parametrize("mode", ["soft_e0", "linear_e0"])
def test_bf16_staged_path_is_bit_identical_to_the_one_call_path(E, mode, golden_dir):
    """hx_hirl_critic_grads + hx_adam + hx_hirl_actor_backward + hx_hirl_actor_wgrad + hx_adam (what a sharded run calls between its
    exchanges; adam_kernel maintains the images) == hx_hirl_learn (the optimizer steps inside the weight-gradient launches)."""
    a, la = run_mode(E, mode, golden_dir, staged=False, collect=[])
    b, lb = run_mode(E, mode, golden_dir, staged=True, collect=[])
    np.testing.assert_allclose(la, lb, rtol=1e-6, atol=1e-7)  # (loss sums are float atomics: order-dependent in the last bit)
    for x, y in ((a.actor, b.actor), (a.critic, b.critic), (a.target_actor, b.target_actor), (a.target_critic, b.target_critic),
                 (a.m_actor, b.m_actor), (a.v_critic, b.v_critic), (a.images.view(torch.int16), b.images.view(torch.int16))):
        assert torch.equal(x, y)


def test_bf16_bc_pretraining_and_td3(E):
    """hx_bc_train_actor and the TD3 (LeakyReLU, no BC) instantiations of the bf16 kernels against the rounded-operand oracle."""
    params, data = D.make_params(11), D.make_data(12)
    ring, exp, bc = device_tables(data)
    rng = np.random.default_rng(3)
    # TD3
    e = E.HirlEngine(batch=128, slope=0.01, use_bc=False)
    e.load_params(params["actor"], params["critic"])
    e.set_update_dtype("bf16")
    o = H.HirlOracle(params["actor"], params["critic"], None, slope=0.01, use_bc=False)
    for k in range(4):
        idx = rng.integers(0, D.N_REPLAY, 128).astype(np.int32)
        noise = rng.normal(0, 0.2, 4).astype(np.float32)
        was_actor = e.actor_trainable
        sync_oracle(o, e, E)
        pre = (e.actor.clone(), e.critic.clone())
        e.assemble(ring, torch.from_numpy(idx).cuda())
        e.learn(noise=torch.from_numpy(noise).cuda())
        rows = data["replay"][idx]
        with H.Bf16Layer2(), KernelKinks(learn_masks(e, E, pre, was_actor, soft=False, use_bc=False)):
            ref = o.learn((rows[:, 0:13], rows[:, 13:17], rows[:, 17:30], rows[:, 30], rows[:, 31]), None, noise)
        np.testing.assert_allclose(e.losses_host()[:2], ref[:2], rtol=LOSS_RTOL, atol=LOSS_ATOL)
        check_grads(e.grad_critic, o.last_grads["critic"], E.CRITIC_LAYOUT, f"td3 call {k} critic")
        if was_actor:
            check_grads(e.grad_actor, o.last_grads["actor"], E.ACTOR_LAYOUT, f"td3 call {k} actor")
        assert_images_current(e, E, f"td3 call {k}")
    # BC pre-training (BC.py:160-185)
    e = E.HirlEngine(batch=128, slope=0.01)
    e.load_params(params["actor"], params["critic"], params["bc_actor"])
    e.set_update_dtype("bf16")
    o = H.HirlOracle(params["actor"], params["critic"], params["bc_actor"], slope=0.01)
    for k in range(3):
        ibc = rng.integers(0, D.N_EXPERT, 128).astype(np.int32)
        sync_oracle(o, e, E)
        o.opt_actor.t = e.actor_step
        pre_actor = E.unpack(e.actor.clone(), E.ACTOR_LAYOUT)
        e.assemble(ring, torch.from_numpy(rng.integers(0, D.N_REPLAY, 128).astype(np.int32)).cuda(), bc_table=bc, idx_bc=torch.from_numpy(ibc).cuda())
        e.bc_train_actor()
        m0, m1 = kernel_masks(e, "ABC", pre_actor["layernorm2.weight"].cpu(), pre_actor["layernorm2.bias"].cpu())
        with H.Bf16Layer2(), KernelKinks({0: m0, 1: m1}):
            loss = H.bc_train_actor(o, (data["expert_s"][ibc], data["expert_a"][ibc]))
        np.testing.assert_allclose(e.losses_host()[2], loss, rtol=LOSS_RTOL)
        check_grads(e.grad_actor, o.last_grads["actor"], E.ACTOR_LAYOUT, f"bc step {k}")
        assert_images_current(e, E, f"bc step {k}")


def test_fp32_update_is_untouched_by_the_bf16_machinery(E, golden_dir):
    """An engine switched to the bf16 update and back computes, in fp32 mode, bit for bit what an engine that never left it computes."""
    g = np.load(os.path.join(golden_dir, "hirl_learn_soft_e0.npz"))
    params, data = D.make_params(D.PARAM_SEED), D.make_data(D.DATA_SEED)
    ring, exp, bc = device_tables(data)
    engines = []
    for toggle in (False, True):
        e = E.HirlEngine(batch=128)
        e.load_params(params["actor"], params["critic"], params["bc_actor"])
        if toggle:
            e.set_update_dtype("bf16")
            e.set_update_dtype("f32")
        for k in range(6):
            e.assemble(ring, torch.from_numpy(g["idx_buf"][k].astype(np.int32)).cuda(), bc_table=bc, idx_bc=torch.from_numpy(g["idx_bc"][k].astype(np.int32)).cuda())
            e.learn(noise=torch.from_numpy(g["noise"][k]).cuda(), bc_weight_now=100 if g["bc_w_in"][k] == 100 else float(g["bc_w_in"][k]))
        engines.append(e)
    a, b = engines
    np.testing.assert_allclose(a.losses_host(), b.losses_host(), rtol=1e-6, atol=1e-7)  # (loss sums are float atomics: last-bit order effects)
    for x, y in ((a.actor, b.actor), (a.critic, b.critic), (a.target_actor, b.target_actor), (a.target_critic, b.target_critic)):
        assert torch.equal(x, y)
    # the bf16 update differs from the fp32 one (it IS another arithmetic), modestly
    c = E.HirlEngine(batch=128)
    c.load_params(params["actor"], params["critic"], params["bc_actor"])
    c.set_update_dtype("bf16")
    c.assemble(ring, torch.from_numpy(g["idx_buf"][0].astype(np.int32)).cuda(), bc_table=bc, idx_bc=torch.from_numpy(g["idx_bc"][0].astype(np.int32)).cuda())
    c.learn(noise=torch.from_numpy(g["noise"][0]).cuda(), bc_weight_now=100)
    l16, l32 = c.losses_host()[0], float(g["out"][0][0])
    assert l16 != l32 and abs(l16 - l32) <= 5e-2 * abs(l32)


@pytest.mark.parametrize("mode", ["soft_e0", "linear_e0"])
def test_bf16_free_running_against_the_fp32_reference_recording(E, mode, golden_dir):
    """10 consecutive learn() calls of the bf16 update path with NO re-synchronisation to any oracle (the per-call tests above re-seat the oracle
    on the engine's state before every call and import the kernel's ReLU masks): the engine's own trajectory against the REFERENCE's recorded
    fp32 run (tests/golden/hirl_learn_*.npz: losses and 128 probed entries per network after every call).
    Stated bounds (tools/ubench/bf16_free_run.py prints the distributions; the bars are ~3x what this seed shows: after the 10th call the
    probed actor / critic entries differ by 2-7e-6 in the median, 99.2 % are within 2e-4, the largest is 2.8e-4; targets 9e-7):
      * critic loss within 2 % at every call, BC loss within 1 %, the soft / linear BC weight within 0.02 absolute;
      * actor and critic after every call: median of the 128 probed entries' |difference| < 2e-5, >= 97 % within 2e-4, none beyond 1e-3
        (Adam turns a bf16-level difference of a resolved gradient entry into ~1e-6 per step, a sign flip of a near-zero entry into up to
        2 lr = 2e-3 per step: the tail is the sign flips);
      * targets (Polyak, tau 0.005): every probed entry within 3e-6."""
    g = np.load(os.path.join(golden_dir, f"hirl_learn_{mode}.npz"))
    params, data = D.make_params(D.PARAM_SEED), D.make_data(D.DATA_SEED)
    ring, exp, bc = device_tables(data)
    e = E.HirlEngine(batch=128)
    e.load_params(params["actor"], params["critic"], params["bc_actor"])
    e.set_update_dtype("bf16")
    from tests.test_hirl_gpu import probes_of

    stats = []
    for k in range(g["out"].shape[0]):
        idx = np.concatenate([g["idx_buf"][k], g["idx_exp"][k]]).astype(np.int32)
        w_in = 100 if g["bc_w_in"][k] == 100 else float(g["bc_w_in"][k])
        e.assemble(ring, torch.from_numpy(idx).cuda(), expert_ring=exp, n_main=128 - int(g["expert_num"]), bc_table=bc,
                   idx_bc=torch.from_numpy(g["idx_bc"][k].astype(np.int32)).cuda())
        e.learn(noise=torch.from_numpy(g["noise"][k]).cuda(), bc_weight_now=w_in, bc_warm_up_weight=float(g["warm_in"][k]))
        got, ref = np.asarray(e.losses_host()), g["out"][k]
        assert abs(got[0] - ref[0]) <= 2e-2 * abs(ref[0]), f"{mode} call {k}: critic loss {got[0]} vs {ref[0]}"
        if k % 2 == 0:  # actor calls: bc loss and weight
            assert abs(got[2] - ref[2]) <= 1e-2 * abs(ref[2]) + 1e-6, f"{mode} call {k}: bc loss {got[2]} vs {ref[2]}"
            assert abs(got[5] - ref[5]) <= 0.02, f"{mode} call {k}: bc weight {got[5]} vs {ref[5]}"
        for j, (s, a, v) in enumerate(probes_of(e, E)):
            d = np.abs(v - g["probe_val"][k][j])
            stats.append((k, j, float(np.median(d)), float((d <= 2e-4).mean()), float(d.max())))
            if j in (0, 1):  # actor, critic
                assert np.median(d) < 2e-5 and (d <= 2e-4).mean() >= 0.97 and d.max() <= 1e-3, (mode, k, j, stats[-1])
            else:            # targets
                assert d.max() <= 3e-6, (mode, k, j, stats[-1])
    if os.environ.get("HX_PRINT_STATS"):
        for s in stats:
            print("call %d net %d: median %.2e  within 2e-4: %.3f  max %.2e" % s)
