"""The FRONT launch (hx_hirl_front + hx_hirl_learn_back, HirlEngine.step_learn): the env step of a vector loop and the first two launches of the
learn() call behind it as ONE launch, the target critics waiting in-launch for the target actor's rows.

Parity statement: one step_learn == act_step, then learn() on a minibatch drawn with HxSample.total = the ring's total BEFORE the env step and
HxSample.guard = n (the slots the step may overwrite left out), launch B in the front launch's tiling (hx_debug_set_fwd_nt) — bit for bit in everything both leave behind: actions, env state words,
observations, rewards, episode counters, the replay rows (as a multiset: ring slots are handed out by an atomic), drawn indices, smoothing noise,
row tiles, networks, Adam moments; loss sums to 1e-6 (they are accumulated with atomics).  Checked step by step from shared states over rings that
fill up and wrap, HIRL (soft / fixed weights, BC minibatch, expert rows) and TD3, ReLU and leaky networks, every acting role of the launch: per-tile 32-row
workgroups (fp32 MFMA, exact split, bf16), the persistent weight-stationary bf16 kernel (beyond 4,096 envs; on every CU beyond 32,768) and the streaming
exact-split kernel (from 8,192 envs on), one-call and sharded (staged) update sequences.  The draw rule itself is checked against a restatement on the host
(allowed slots, no replacement), the loop against the env and update oracles directly."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from tests import _hirl_data as D  # noqa: E402

NETS = ("actor", "critic", "target_actor", "target_critic", "m_actor", "v_actor", "m_critic", "v_critic")


@pytest.fixture(scope="module")
def eng_mod():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from hirl4ucav_amd.agents import engine

    return engine


def allowed_slots(tot, cap, guard):
    """The population of a guarded draw, restated: every slot that holds a transition before the env step (slots < min(tot, cap)) and is not one of
    the `guard` slots behind the ring head tot % cap."""
    live = np.arange(min(tot, cap))
    window = (tot + np.arange(guard)) % cap
    return np.setdiff1d(live, window)


def make_pair(eng_mod, n, cap, use_bc, slope, seed, act="f32", staged=False):
    from hirl4ucav_amd.environments.batched import BatchedHarfangEnv
    from hirl4ucav_amd.utils.buffer import DeviceReplay

    params = D.make_params(31)
    rng = np.random.default_rng(seed)
    exp = DeviceReplay(64)
    exp.store_rows(torch.from_numpy(rng.normal(size=(40, 32)).astype(np.float32)))
    bc = torch.from_numpy(rng.normal(size=(150, 32)).astype(np.float32)).cuda() if use_bc else None
    scen = (np.arange(n) % 3).astype(np.int32)
    side = []
    for _ in range(2):
        e = eng_mod.HirlEngine(batch=128, use_bc=use_bc, slope=slope)
        e.x9_rows, e.front_x9 = None, False  # "f32" = fp32 MFMA in the front launch too, so that act_step (16-row workgroups, fp32 MFMA below x9_rows) is its reference
        e.staged = staged  # the sharded rank's launch sequence (one rank: the exchanges are no-ops)
        e.load_params(params["actor"], params["critic"], params["bc_actor"] if use_bc else None)
        if act == "bf16":  # the bf16 update path with the bf16 acting format
            e.set_update_dtype("bf16")
        if act != "f32":
            e.set_act_dtype(act)
        rep = DeviceReplay(cap)
        env = BatchedHarfangEnv(n, scenario=scen, seed=5, max_step=6, auto_reset=True, random_reset=True, replay=rep)  # (max_step 6: time-limit steps are not stored, so a step inserts fewer than n rows)
        env.reset()
        side.append((e, env, rep))
    return side, exp, bc


def sync(dst, src):
    (e1, env1, rep1), (e0, env0, rep0) = dst, src
    e1.arena.copy_(e0.arena)
    for k in ("critic_step", "actor_step", "update_count", "actor_trainable", "sample_calls", "act_calls"):
        setattr(e1, k, getattr(e0, k))
    env1._state_store.copy_(env0._state_store)
    for k in ("obs", "reward", "done", "success", "episode_ctr"):
        getattr(env1, k).copy_(getattr(env0, k))
    if env0.stats is not None:
        env1.stats.copy_(env0.stats)
    rep1.ring.copy_(rep0.ring); rep1.success.copy_(rep0.success); rep1.total.copy_(rep0.total)


def sorted_rows(rep):
    r = rep.ring.cpu().numpy()
    return r[np.lexsort(r.T[::-1])]


@pytest.mark.parametrize("use_bc,slope,n,cap,act,staged", [(True, 0.0, 1024, 2600, "f32", False), (False, 0.01, 576, 1400, "f32", False), (True, 0.0, 4096, 10000, "f32", False),
                                                           (True, 0.0, 4096, 10000, "f32x9", False), (False, 0.01, 1000, 2400, "f32x9", False),
                                                           (True, 0.0, 1024, 2600, "f32", True), (False, 0.0, 2048, 5000, "f32x9", True),
                                                           (True, 0.0, 4096, 10000, "bf16", False), (False, 0.01, 1000, 2400, "bf16", True),
                                                           (True, 0.0, 16384, 40000, "bf16", False), (False, 0.01, 12000, 30000, "bf16", True),
                                                           (True, 0.0, 8192, 20000, "bf16", False), (True, 0.0, 5000, 12000, "bf16", False),  # bf16 beyond 4,096 envs: persistent acting workgroups
                                                           (True, 0.0, 40000, 90000, "bf16", False),  # ... beyond 32,768: on every CU, the update's workgroups behind them
                                                           (True, 0.0, 8192, 20000, "f32x9", False), (False, 0.01, 12288, 30000, "f32x9", True), (True, 0.0, 10000, 24000, "f32x9", False),  # exact split from 8,192 envs on: one 64-row pass per acting workgroup ...
                                                           (True, 0.0, 20000, 44000, "f32x9", False)])  # ... beyond 16,384: passes over every CU, the update's workgroups behind them
def test_front_launch_equals_act_step_then_guarded_learn(eng_mod, use_bc, slope, n, cap, act, staged):
    from hirl4ucav_amd import _lib
    L = _lib.load()
    side, exp, bc = make_pair(eng_mod, n, cap, use_bc, slope, seed=n, act=act, staged=staged)
    (a, env_a, rep_a), (b, env_b, rep_b) = side
    a.act_step(env_a, sigma=0.1, seed=3)  # some rows in the ring before the first draw
    snap = torch.zeros(1, dtype=torch.int64, device="cuda")
    seen_wrapped_window = False
    for k in range(16):
        sync(side[1], side[0])
        w = ((100 if k % 4 == 0 else (None if k % 4 < 3 else 0.3)) if use_bc else 0.0)
        tot0 = int(rep_a.total.item())
        out_a = a.step_learn(env_a, exp, bc, n_main=96, act_sigma=0.1, act_seed=3, sample_seed=11, bc_weight_now=w, bc_warm_up_weight=0.05)
        # the same step as separate launches: total before the step, guard = n
        snap.copy_(rep_b.total)
        out_b = b.act_step(env_b, sigma=0.1, seed=3)
        b.sample(rep_b, exp, bc, n_main=96, seed=11, defer=True)
        b._pending[0].total, b._pending[0].guard = snap.data_ptr(), n
        L.hx_debug_set_fwd_nt(64, 1, 1)  # launch B (the second forward launch of learn()) in the front launch's tiling of launch B (64-column workgroups whatever the job count: same K-split, same bits)
        try:
            b.learn(bc_weight_now=w, bc_warm_up_weight=0.05)
        finally:
            L.hx_debug_set_fwd_nt(0, 0, 0)
        a.front_check()
        for x, y, name in zip(out_a, out_b, ("actions", "obs", "reward", "done", "success")):
            assert torch.equal(x, y), (k, name)
        assert torch.equal(env_a._state_store, env_b._state_store) and torch.equal(env_a.episode_ctr, env_b.episode_ctr), k
        assert torch.equal(rep_a.total, rep_b.total), k
        np.testing.assert_array_equal(sorted_rows(rep_a), sorted_rows(rep_b), err_msg=f"step {k}: replay rows")
        for name in ("_idx", "_noise", "rows") + (("_idx_bc", "bc_rows") if use_bc else ()):
            assert torch.equal(getattr(a, name), getattr(b, name)), (k, name)
        np.testing.assert_allclose(a.losses_host(), b.losses_host(), rtol=1e-6, atol=1e-7, err_msg=f"step {k}")
        for name in NETS:
            assert torch.equal(getattr(a, name), getattr(b, name)), (k, name)
        # the draw rule on the host
        i = a._idx.cpu().numpy()
        ok = allowed_slots(tot0, cap, n)
        assert len(set(i[:96])) == 96 and np.isin(i[:96], ok).all(), (k, tot0)
        assert len(set(i[96:])) == 32 and i[96:].max() < 40
        if tot0 >= cap and tot0 % cap + n > cap:
            seen_wrapped_window = True
        # the next call's minibatch is in waiting (drawn inside this call's learn() part), and only the first call drew with a launch of its own
        assert a._front_drawn[:2] == (env_a, env_a.steps_issued) and a._front_drawn[8] == a.sample_calls + 1
    assert int(rep_a.total.item()) > cap, "the ring was meant to wrap"
    assert seen_wrapped_window, "no step had its guard window across the ring's end"
    assert a.update_count == 8


def test_guarded_draw_covers_its_population(eng_mod):
    """Every allowed slot is drawn and no other: 200 guarded draws of 128 from a small ring, full and not, with the window wrapped and not."""
    from hirl4ucav_amd.utils.buffer import DeviceReplay

    params = D.make_params(31)
    rng = np.random.default_rng(3)
    cap, guard = 700, 150
    rep = DeviceReplay(cap)
    rep.ring.copy_(torch.from_numpy(rng.normal(size=(cap, 32)).astype(np.float32)))
    e = eng_mod.HirlEngine(batch=128, use_bc=False)
    e.load_params(params["actor"], params["critic"], None)
    for tot in (400, 650, 700, 2 * 700 + 10, 3 * 700 + 620):  # not full; window past the end; just full; full; full with the window wrapped
        rep.total.fill_(tot)
        seen = set()
        ok = allowed_slots(tot, cap, guard)
        for k in range(200):
            e.sample(rep, n_main=128, seed=7, defer=True)
            e._pending[0].guard = guard
            e.learn()
            i = e._idx.cpu().numpy()
            assert len(set(i)) == 128 and np.isin(i, ok).all(), (tot, k)
            seen.update(i.tolist())
        assert seen == set(ok.tolist()), (tot, len(seen), len(ok))


def test_step_learn_refuses_what_it_does_not_cover(eng_mod):
    side, exp, bc = make_pair(eng_mod, 64, 1024, True, 0.0, seed=1)
    e, env, rep = side[0]
    e.set_act_dtype("bf16")
    with pytest.raises(Exception, match="front launch"):
        e.step_learn(env, exp, bc, n_main=96)
    e.set_act_dtype("f32")
    e.sample(rep, exp, bc, n_main=96, defer=True)
    with pytest.raises(Exception, match="pending"):
        e.step_learn(env, exp, bc, n_main=96)
    e._pending = None
    from hirl4ucav_amd.environments.batched import BatchedHarfangEnv
    from hirl4ucav_amd.utils.buffer import DeviceReplay
    small = DeviceReplay(600)  # fewer than 2 n slots: nothing would be left to draw from once the ring is full
    env2 = BatchedHarfangEnv(512, scenario="straight_line", seed=1, replay=small)
    env2.reset()
    with pytest.raises(Exception, match="half the ring"):
        e.step_learn(env2, exp, bc, n_main=96)


def test_front_launch_default_acting_format_is_the_exact_split(eng_mod):
    """act_dtype "f32" with the engine's defaults: the front launch multiplies in the exact 9-term bf16 split (front_x9), i.e. it equals the front launch
    of an engine set to "f32x9" bit for bit, and its actions stay within 2e-6 of the fp32-MFMA front launch."""
    outs = []
    for cfg in ("default", "f32x9", "mfma"):
        side, exp, bc = make_pair(eng_mod, 1024, 4096, True, 0.0, seed=4, act="f32x9" if cfg == "f32x9" else "f32")
        e, env, rep = side[0]
        if cfg == "default":
            e.front_x9 = True
        env.step(torch.from_numpy(np.random.default_rng(2).uniform(-1, 1, (1024, 4)).astype(np.float32)).cuda())  # the same first step for all three
        out = e.step_learn(env, exp, bc, n_main=96, act_sigma=0.1, act_seed=3, sample_seed=11, bc_weight_now=100)[0]
        outs.append(out.clone())
    assert torch.equal(outs[0], outs[1])
    assert float((outs[0] - outs[2]).abs().max()) < 2e-6 and not torch.equal(outs[0], outs[2])


def test_launch_c_rides_only_where_the_cus_have_time_for_it(eng_mod):
    """HirlEngine.front_c_for: OFF by default since round 5 (the one size class where launch C inside the front launch paid — 8,192 envs fp32 — reads 62.1 -> 61.4 us
    with the six-term acting format, profiles/r05_front_c_8192.txt); "auto" = round 4's rule: only in the streaming acting role around 8,192 envs (fp32, exact
    split) — measured slower everywhere else (profiles/archive/r04c_front_c_ab.txt); True / False force it.  (That the results do not depend on it:
    test_launch_c_inside_the_front_launch_is_bit_identical forces it for every acting role.)"""
    e = eng_mod.HirlEngine(batch=128, use_bc=True)
    assert e.front_c is False and not e.front_c_for(8192, True, 0, False)
    e.front_c = "auto"
    for actor_phase in (False, True):
        assert e.front_c_for(8192, actor_phase, 0, False) and e.front_c_for(8256, actor_phase, 0, False)
        for n in (32, 4096, 6144, 8704, 12288, 16384, 65536):
            assert not e.front_c_for(n, actor_phase, 0, False), n
        assert not e.front_c_for(8192, actor_phase, 0, True)  # the bf16 update path: persistent bf16 acting role
    e.front_c = True
    assert e.front_c_for(4096, False, 0, False)
    e.front_c = False
    assert not e.front_c_for(8192, True, 0, False)


def test_front_loop_free_running(eng_mod):
    """40 free-running steps of the front loop (no re-synchronisation, every minibatch but the first drawn by the previous call's learn() part): finite
    losses, every Adam step counted, the ring wrapped, the hand-off status word clear."""
    side, exp, bc = make_pair(eng_mod, 512, 8192, True, 0.0, seed=9)
    (a, env_a, rep_a), _ = side
    a.act_step(env_a, sigma=0.1, seed=3)
    for k in range(40):
        a.step_learn(env_a, exp, bc, n_main=96, act_sigma=0.1, act_seed=3, sample_seed=11, bc_weight_now=100 if k % 4 == 0 else None, bc_warm_up_weight=0.05)
        assert a._front_drawn[:2] == (env_a, env_a.steps_issued)
        if k == 10:  # the hand-off counters start over long before 32 bits run out: a reset between two launches changes nothing
            a._front_epoch = 400_000_000
        if k == 20:  # an env step from outside: the tiles in waiting no longer fit, the next call draws afresh (a launch of its own) — and goes on
            a.act_step(env_a, sigma=0.1, seed=3)
    a.front_check()
    assert np.isfinite(a.losses_host()).all() and a.critic_step == 40 and a.update_count == 20 and a._front_epoch == 29
    assert int(rep_a.total.item()) > 8192


@pytest.mark.parametrize("n", [32, 768, 4096])
def test_front_loop_against_the_oracles(eng_mod, n):
    """The front loop checked against the CHECKERS directly, step by step (not only against the separate launches): its env step == the C env oracle
    stepped with the actions the launch chose (done / success flags exact, observations and rewards 1e-5); its learn() == the update oracle (pinned to
    the reference's Agent.learn) on the minibatch the loop drew, from synchronised states: losses 2e-5 and — with 32 envs, ONE acting workgroup, where the
    order of the replay rows and with it the whole run is reproducible — every parameter within check_params' bars.  n = 4096 is the HEADLINE's shape
    (BASELINE.json configs[1]: 128 acting workgroups of 32 rows in the exact-split format, launches A and B beside them): it meets both oracles in one hop,
    not only through the bit-identity with the separate launches.  (With several acting workgroups the
    ring slots are handed out by an atomic, every run draws other minibatches, and one in ~500 of them holds a parameter whose gradient sits on a ReLU kink:
    tools/ubench/front_oracle_stress.py shows the separate launches missing the bar on exactly the same entries, bit for bit.)"""
    from oracle import hirl_oracle as H
    from tests import _oracle as ox
    from tests.test_hirl_gpu import assert_losses, check_params, sync_oracle
    from hirl4ucav_amd.environments.batched import BatchedHarfangEnv
    from hirl4ucav_amd.utils.buffer import DeviceReplay

    params, data = D.make_params(D.PARAM_SEED), D.make_data(D.DATA_SEED)
    e = eng_mod.HirlEngine(batch=128)
    e.load_params(params["actor"], params["critic"], params["bc_actor"])
    o = H.HirlOracle(params["actor"], params["critic"], params["bc_actor"])
    bc = np.zeros((data["expert_s"].shape[0], 32), np.float32)
    bc[:, 0:13], bc[:, 13:17] = data["expert_s"], data["expert_a"]
    bc_t = torch.from_numpy(bc).cuda()
    exp = DeviceReplay(D.N_EXPERT)
    exp.store_rows(torch.from_numpy(data["expert_rows"]))
    rep = DeviceReplay(max(4096, 4 * n))
    env = BatchedHarfangEnv(n, scenario="straight_line", seed=1, auto_reset=False, replay=rep)
    env.reset()
    envs, oobs = ox.reset_batch(n, 0, 1, seed=1)
    for pre in range(1 if n > 128 else 6):  # at least 96 + n rows in the ring before the first draw
        a0 = np.random.default_rng(pre).uniform(-1, 1, (n, 4)).astype(np.float32)
        env.step(torch.from_numpy(a0).cuda())
        ox.step_batch(envs, a0, oobs)
    for k in range(8):
        sync_oracle(o, e, eng_mod)
        was_actor = e.actor_trainable
        w = 100 if k % 4 == 0 else (None if k % 4 < 3 else 0.3)
        if w is None:
            o.bc_weight = float(e.wstate.item())  # "keep the stored weight": the oracle keeps its own copy of it
        acts, obs, r, d, sc = e.step_learn(env, exp, bc_t, n_main=96, act_sigma=0.1, act_seed=3, sample_seed=11, bc_weight_now=w, bc_warm_up_weight=0.05)
        ro, do, so = ox.step_batch(envs, acts.cpu().numpy(), oobs)
        np.testing.assert_array_equal(d.cpu().numpy(), do)
        np.testing.assert_array_equal(sc.cpu().numpy(), so)
        np.testing.assert_allclose(obs.cpu().numpy(), oobs, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(r.cpu().numpy(), ro, rtol=1e-5, atol=1e-6)
        rows, bcr = e.rows.cpu().numpy().reshape(128, 32), e.bc_rows.cpu().numpy().reshape(128, 32)
        ref = o.learn((rows[:, 0:13], rows[:, 13:17], rows[:, 17:30], rows[:, 30], rows[:, 31]), (bcr[:, 0:13], bcr[:, 13:17]), e._noise.cpu().numpy(),
                      w if w is not None else o.bc_weight, 0.05)
        assert_losses(e.losses_host(), ref, f"front step {k}")
        if n <= 32:
            check_params(e, o, eng_mod, f"front step {k}", was_actor_call=was_actor)
    e.front_check()


@pytest.mark.parametrize("n,cap,esac", [(16384, 40000, False), (9000, 20000, True)])
def test_sac_front_launch_equals_act_step_then_guarded_learn(n, cap, esac):
    """The SAC front launch (hx_sac_front + hx_sac_learn_back, SacEngine.step_learn): explore + env step + insert and the first forward launch of
    SacAgent.learn in ONE launch (the persistent acting kernel beyond 8,192 envs: exact split from 16,384 rows on, fp32 MFMA below), the minibatch pre-drawn
    by the previous call == act_step, then learn() on a minibatch drawn with HxSample.total read before the step and guard = n — bit for bit, step by step
    from shared states over a ring that fills and wraps; SAC and E-SAC (expert rows mixed in)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from hirl4ucav_amd.agents import sac_engine as SE
    from hirl4ucav_amd.environments.batched import BatchedHarfangEnv
    from hirl4ucav_amd.utils.buffer import DeviceReplay
    from tests.test_oracle_sac import sac_params

    p = sac_params()
    rng = np.random.default_rng(n)
    exp = None
    if esac:
        exp = DeviceReplay(64)
        exp.store_rows(torch.from_numpy(rng.normal(size=(40, 32)).astype(np.float32)))
    scen = (np.arange(n) % 3).astype(np.int32)
    side = []
    for _ in range(2):
        e = SE.SacEngine(batch=128)
        e.load_params(p["policy"], p["q1"], p["q2"])
        rep = DeviceReplay(cap)
        env = BatchedHarfangEnv(n, scenario=scen, seed=5, max_step=6, auto_reset=True, random_reset=True, replay=rep)
        env.reset()
        side.append((e, env, rep))
    (a, env_a, rep_a), (b, env_b, rep_b) = side
    a.act_step(env_a, seed=3)
    snap = torch.zeros(1, dtype=torch.int64, device="cuda")
    for k in range(8):
        b.arena.copy_(a.arena)
        for name in ("learning_steps", "sample_calls", "act_calls"):
            setattr(b, name, getattr(a, name))
        env_b._state_store.copy_(env_a._state_store)
        for name in ("obs", "reward", "done", "success", "episode_ctr"):
            getattr(env_b, name).copy_(getattr(env_a, name))
        rep_b.ring.copy_(rep_a.ring); rep_b.success.copy_(rep_a.success); rep_b.total.copy_(rep_a.total)
        tot0 = int(rep_a.total.item())
        out_a = a.step_learn(env_a, exp, n_main=96, act_seed=3, sample_seed=11)
        snap.copy_(rep_b.total)
        out_b = b.act_step(env_b, seed=3)
        b.sample(rep_b, exp, n_main=96, seed=11, defer=True)
        b._pending[0].total, b._pending[0].guard = snap.data_ptr(), n
        b.learn()
        for x, y, name in zip(out_a, out_b, ("actions", "obs", "reward", "done", "success")):
            assert torch.equal(x, y), (k, name)
        assert torch.equal(env_a._state_store, env_b._state_store) and torch.equal(rep_a.total, rep_b.total), k
        np.testing.assert_array_equal(sorted_rows(rep_a), sorted_rows(rep_b), err_msg=f"step {k}: replay rows")
        assert torch.equal(a._idx, b._idx) and torch.equal(a.rows, b.rows), k
        np.testing.assert_allclose(a.losses_host(), b.losses_host(), rtol=1e-6, atol=1e-7, err_msg=f"step {k}")
        for name in ("policy", "critic", "target_critic", "m_policy", "v_policy", "m_critic", "v_critic", "alpha_state"):
            assert torch.equal(getattr(a, name), getattr(b, name)), (k, name)
        i = a._idx.cpu().numpy()
        m = 96 if esac else 128
        assert len(set(i[:m])) == m and np.isin(i[:m], allowed_slots(tot0, cap, n)).all(), (k, tot0)
    assert int(rep_a.total.item()) > cap



@pytest.mark.gpu
def test_launch_c_inside_the_front_launch_is_bit_identical():
    """HX_FRONT_C=1 (tuning knob, read once per process: hence the child): launch C — the critics' backward — rides in the front launch, its workgroups
    waiting in-launch for launches A and B (hx_bwd_body.h, FRONT == 3).  The same parity statement must hold for every acting role, one-call and staged."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from hirl4ucav_amd.agents import engine\n"
            "from tests import test_front_gpu as T\n"
            "fn = T.test_front_launch_equals_act_step_then_guarded_learn\n"
            "for case in [(True, 0.0, 4096, 10000, 'f32x9', False), (False, 0.01, 576, 1400, 'f32', False), (True, 0.0, 1024, 2600, 'f32', True),\n"
            "             (True, 0.0, 4096, 10000, 'bf16', False), (False, 0.01, 12000, 30000, 'bf16', True), (True, 0.0, 8192, 20000, 'f32x9', False),\n"
            "             (True, 0.0, 40000, 90000, 'bf16', False)]:\n"
            "    fn(engine, *case)\n"
            "print('front-c ok')\n") % root
    env = dict(os.environ, HX_FRONT_C="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "front-c ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
