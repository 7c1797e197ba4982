"""GPU parity tests of the env-step path (hx_env_* through the C ABI) against the CPU oracle and the committed
golden fixtures.  Masks / integer state: bit-exact.  fp32 state words, observation and reward: ALSO bit-exact — model v2
(docs/DYNAMICS.md) builds everything, the read-back's inverse trigonometry included, from correctly rounded operations
(+ - * / sqrt fma), which is stricter than north_star's "fp32 dynamics within 1e-5 rel".  Comparisons with the REFERENCE's
float64 outputs (golden fixtures) keep rtol 1e-5."""
import os

import numpy as np
import pytest

from tests import _oracle as ox

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-6


@pytest.fixture(scope="module")
def hx():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from hirl4ucav_amd import _lib
    from hirl4ucav_amd.environments.batched import BatchedHarfangEnv
    from hirl4ucav_amd.utils.buffer import DeviceReplay

    _lib.load()  # raises if the HIP extension is missing: no silent fallback

    class NS:
        pass

    ns = NS()
    ns.lib, ns.Env, ns.Replay = _lib, BatchedHarfangEnv, DeviceReplay
    return ns


def random_states(n, rng, scen):
    """Plausible but adversarial env states (AoS [n, 37], oracle layout) that hit every branch of the step."""
    e = np.zeros((n, 37), np.float32)
    u = e.view(np.uint32)

    def unit(k):
        v = rng.normal(size=(n, k))
        return v / np.linalg.norm(v, axis=1, keepdims=True)

    e[:, 0:3] = rng.normal(0, 1500, (n, 3)) + np.array([0, 4000, -2000])
    e[:, 13:16] = rng.normal(0, 1500, (n, 3)) + np.array([0, 4200, 0])
    # some pairs close together (inside lock range, inside the cone)
    close = rng.random(n) < 0.4
    e[close, 13:16] = e[close, 0:3] + unit(3)[close] * rng.uniform(50, 3500, (close.sum(), 1))
    # altitude edge cases around 500 / 2000 / 7000 / 10000
    edges = np.array([499.99, 500.0, 500.01, 1999.9, 2000.0, 2000.1, 6999.9, 7000.0, 7000.1, 9999.9, 10000.0, 10000.1], np.float32)
    k = rng.random(n) < 0.1
    e[k, 1] = rng.choice(edges, k.sum())
    e[:, 3:6] = unit(3) * rng.uniform(80, 450, (n, 1))
    e[:, 16:19] = unit(3) * rng.uniform(80, 450, (n, 1))
    e[:, 6:10] = unit(4)
    e[:, 19:23] = unit(4)
    # often point the ally roughly at the opponent so that the cone test is exercised near its boundary
    aim = rng.random(n) < 0.5
    d = (e[:, 13:16] - e[:, 0:3]).astype(np.float64)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d += rng.normal(0, 0.2, (n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    # quaternion rotating +Z onto d
    z = np.array([0, 0, 1.0])
    ax = np.cross(np.broadcast_to(z, d.shape), d)
    w = 1.0 + d[:, 2]
    q = np.concatenate([w[:, None], ax], 1)
    q /= np.linalg.norm(q, axis=1, keepdims=True) + 1e-12
    e[aim, 6:10] = q[aim]
    e[:, 10:13] = rng.uniform(-1, 1, (n, 3))
    e[:, 23:26] = rng.uniform(-0.3, 0.3, (n, 3))
    e[:, 29:32] = unit(3) * rng.uniform(100, 1000, (n, 1))
    e[:, 26:29] = e[:, 13:16] + unit(3) * rng.uniform(5, 400, (n, 1))  # missile near the opponent: hits happen
    e[:, 32] = rng.choice(np.array([0.2, 0.2, 0.2, 0.1, np.nextafter(np.float32(0.1), np.float32(0)), 0.05, 0.0], np.float32), n)
    e[:, 33] = rng.uniform(0, 1.2, n)
    e[:, 33][rng.random(n) < 0.2] = np.float32(1.0) - np.float32(1 / 60)  # on the verge of locking
    e[:, 34] = rng.uniform(0, 20.5, n)
    bits = rng.integers(0, 2, (n, 15)).astype(np.uint32)
    flags = np.zeros(n, np.uint32)
    for b in (0, 1, 2, 3, 4, 5, 10, 11, 12, 13, 14):
        flags |= bits[:, b] << np.uint32(b)
    flags |= (np.asarray(scen, np.uint32) << np.uint32(8))
    u[:, 35] = flags
    ep = rng.integers(0, 60, n).astype(np.uint32)
    script = rng.integers(0, 520, n).astype(np.uint32)
    script[rng.random(n) < 0.2] = rng.choice(np.array([98, 99, 100, 248, 249, 250, 498, 499, 500], np.uint32), 1)
    u[:, 36] = ep | (script << np.uint32(16))
    return e


def to_soa(e):
    return torch.from_numpy(np.ascontiguousarray(e.T))


def from_soa(t):
    return np.ascontiguousarray(t.cpu().numpy().T)


def assert_float_close(a, b, what):
    """fp32 outputs of the kernel vs the oracle: the same bit patterns (NaN-free by construction)."""
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    bad = a.view(np.uint32) != b.view(np.uint32)
    assert not bad.any(), f"{what}: {int(bad.sum())} of {a.size} values differ; first {a[bad][:4]} vs {b[bad][:4]}"


def assert_ref_close(a, b, what):
    """kernel vs the REFERENCE wrapper's own (float64) outputs: the tolerance north_star states"""
    np.testing.assert_allclose(a, b, rtol=RTOL, atol=ATOL, err_msg=what)


LAYOUTS = [(0, 0), (1, 32), (1, 64), (1, 128), (1, 256), (1, 512), (0, 64), (0, 128), (0, 256)]


@pytest.mark.parametrize("pair,epb", LAYOUTS[1:])
@pytest.mark.parametrize("n", [1, 31, 4097])
def test_every_launch_shape_gives_the_same_bits(hx, n, pair, epb):
    """hx_env_step picks its launch shape from n (one or two lanes per env, 32..512 envs per workgroup); every shape must
    produce the same state, outputs, replay rows and statistics, bit for bit — with auto-reset, time limit and insert on."""
    rng = np.random.default_rng(7 * n + epb + pair)
    scen = rng.integers(0, 3, n)
    e = random_states(n, rng, scen)
    e.view(np.uint32)[:, 36] = (e.view(np.uint32)[:, 36] & 0xFFFF0000) | rng.integers(40, 50, n).astype(np.uint32)
    obs_prev = rng.uniform(-1, 1, (n, 13)).astype(np.float32)
    outs = []
    for lay in (0, hx.lib.layout(pair, epb)):
        rep = hx.Replay(1 << 15, "cuda")
        env = hx.Env(n, scenario=0, auto_reset=True, max_step=50, random_reset=True, seed=3, replay=rep, layout=lay)
        env.set_state(to_soa(e), torch.from_numpy(obs_prev))
        r2 = np.random.default_rng(5)
        for _ in range(6):
            a = r2.uniform(-1, 1, (n, 4)).astype(np.float32)
            env.step(torch.from_numpy(a).cuda())
        torch.cuda.synchronize()
        k = int(rep.total.item())
        rows = rep.ring[:k].cpu().numpy()
        rows = rows[np.lexsort(rows.T[::-1])]
        outs.append((from_soa(env.state).view(np.uint32), env.obs.cpu().numpy().view(np.uint32), env.reward.cpu().numpy().view(np.uint32),
                     env.done.cpu().numpy(), env.success.cpu().numpy(), rows.view(np.uint32), env.episode_ctr.cpu().numpy(),
                     np.asarray(list(env.stats_dict().values()))))
    for x, y in zip(*outs):
        np.testing.assert_array_equal(x, y)


@pytest.mark.parametrize("n", [1, 63, 200_000])
def test_single_step_parity_on_random_states(hx, n):
    """Identical (state, action) batches through the HIP kernel and the scalar oracle: state words and all masks
    bit-exact, obs / reward within 1e-5 rel.  3 x 200k = 6e5 pairs at the large size + ragged small sizes."""
    rng = np.random.default_rng(n)
    scen = rng.integers(0, 3, n)
    e = random_states(n, rng, scen)
    a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    a[rng.random(n) < 0.05, 3] = 0.0
    obs_prev = rng.uniform(-1, 1, (n, 13)).astype(np.float32)
    env = hx.Env(n, scenario=0, auto_reset=False, max_step=0, collect_stats=True)
    env.set_state(to_soa(e), torch.from_numpy(obs_prev))
    obs, r, d, s = env.step(torch.from_numpy(a).cuda())
    torch.cuda.synchronize()
    eo = e.copy()
    oo = obs_prev.copy()
    stats = np.zeros(9, np.uint64)
    ro, do, so = ox.step_batch(eo, a, oo, stats=stats)
    g = from_soa(env.state)
    np.testing.assert_array_equal(d.cpu().numpy(), do)
    np.testing.assert_array_equal(s.cpu().numpy(), so)
    np.testing.assert_array_equal(g.view(np.uint32)[:, 35:], eo.view(np.uint32)[:, 35:])  # flags, counters
    np.testing.assert_array_equal(g.view(np.uint32)[:, :35], eo.view(np.uint32)[:, :35])  # dynamics: bit-exact
    assert_float_close(obs.cpu().numpy(), oo, "obs")
    assert_float_close(r.cpu().numpy(), ro, "reward")
    st = env.stats_dict()
    assert st["fires"] == int(stats[4]) and st["good_fires"] == int(stats[5]) and st["locked_steps"] == int(stats[6])
    assert st["env_steps"] == n
    if n > 1000:  # the batch really exercised the branches
        assert do.sum() > 100 and (so == 1).sum() > 100 and (so == -1).sum() > 100
        f = eo.view(np.uint32)[:, 35]
        assert (f & ox.F_EPISODE_SUCCESS).astype(bool).sum() > 100 and (ro > 500).sum() > 100


def test_trajectory_with_auto_reset_and_fused_insert(hx):
    """4,096 mixed-scenario envs, 120 steps, time limit 50, auto reset, fused replay insert: every step's outputs,
    the whole state, the ring contents (as a multiset per step: slot order inside one launch is decided by an
    atomic), the stored count and the statistics match the oracle."""
    n, cap, max_step, steps = 4096, 4096 * 40, 50, 120
    scen = np.sort(np.arange(n) % 3).astype(np.int32)  # sorted by scenario (contiguous thirds)
    rng = np.random.default_rng(7)
    rep = hx.Replay(cap)
    env = hx.Env(n, scenario=scen, seed=99, max_step=max_step, auto_reset=True, random_reset=True, env_id0=1000, replay=rep)
    obs = env.reset()
    envs, oobs = ox.reset_batch(n, scen, 1, seed=99, env_id0=1000)
    torch.cuda.synchronize()
    assert_float_close(obs.cpu().numpy(), oobs, "reset obs")
    np.testing.assert_array_equal(from_soa(env.state).view(np.uint32), envs.view(np.uint32))
    ring = np.zeros((cap, 32), np.float32)
    rsucc = np.zeros(cap, np.int8)
    total = np.zeros(1, np.uint64)
    stats = np.zeros(9, np.uint64)
    epi = np.zeros(n, np.uint32)
    for t in range(steps):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        # a pursuit-ish bias so that locks / fires / kills happen inside 120 steps
        a[:, 3] = np.where(rng.random(n) < 0.02, 1.0, -1.0)
        # both sides start every step from the SAME previous observation (the oracle's), so that fp32 rounding of
        # transcendental functions cannot accumulate through the stored rows
        env.obs.copy_(torch.from_numpy(oobs))
        before = int(total[0])
        obs, r, d, s = env.step(torch.from_numpy(a).cuda())
        ro, do, so = ox.step_batch(envs, a, oobs, max_step=max_step, auto_reset=1, randomize=1, seed=99, env_id0=1000,
                                   episode_ctr=epi, ring=ring, ring_succ=rsucc, total=total, stats=stats)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(d.cpu().numpy(), do, err_msg=f"done step {t}")
        np.testing.assert_array_equal(s.cpu().numpy(), so, err_msg=f"success step {t}")
        assert_float_close(r.cpu().numpy(), ro, f"reward step {t}")
        assert_float_close(obs.cpu().numpy(), oobs, f"obs step {t}")
        np.testing.assert_array_equal(from_soa(env.state).view(np.uint32), envs.view(np.uint32), err_msg=f"state step {t}")
        after = int(total[0])
        assert int(rep.total.item()) == after
        if after > before:
            g = rep.ring[before:after].cpu().numpy()
            o = ring[before:after]
            key = lambda x: np.lexsort(x.view(np.uint32).T[::-1])  # noqa: E731
            gs, os_ = g[key(g)], o[key(o)]
            # columns that are copies of inputs (prev obs, action) and masks: exact; new obs / reward: tolerance
            np.testing.assert_array_equal(np.sort(g[:, :17].view(np.uint32), axis=0), np.sort(o[:, :17].view(np.uint32), axis=0))
            assert_float_close(gs[:, 17:31], os_[:, 17:31], f"ring rows step {t}")
            np.testing.assert_array_equal(gs[:, 31], os_[:, 31])
            np.testing.assert_array_equal(np.sort(rep.success[before:after].cpu().numpy()), np.sort(rsucc[before:after]))
        np.testing.assert_array_equal(env.episode_ctr.cpu().numpy().view(np.uint32), epi)
    st = env.stats_dict()
    for k, name in enumerate(("episodes", "kills", "fire_success_episodes", "time_limit", "fires", "good_fires", "locked_steps", "env_steps")):
        assert st[name] == int(stats[k]), name
    assert st["episodes"] >= 2 * n and st["time_limit"] >= 2 * n - 10 and st["fires"] > 1000
    # the step that reaches max_step is executed but not stored (train_all.py:346-347)
    assert int(total[0]) <= n * steps - 2 * n + 50


@pytest.mark.parametrize("tag", ["straight_line", "serpentine", "circular"])
def test_closed_loop_golden_on_gpu(hx, tag, golden_dir):
    """Replay the reference-generated closed-loop traces (reference wrapper over the oracle simulator) on the GPU."""
    g = np.load(os.path.join(golden_dir, f"env_closedloop_{tag}.npz"))
    env = hx.Env(1, scenario=int(g["scenario"]), auto_reset=False, random_reset=False)
    obs = env.reset()
    assert_ref_close(obs.cpu().numpy()[0], g["obs0"], "obs0")
    acts = torch.from_numpy(g["actions"]).cuda()
    for t in range(acts.shape[0]):
        obs, r, d, s = env.step(acts[t:t + 1].contiguous())
        assert_ref_close(obs.cpu().numpy()[0], g["obs"][t], f"obs {t}")
        assert_ref_close(r.item(), g["reward"][t], f"reward {t}")
        assert (int(d.item()), int(s.item())) == (int(g["done"][t]), int(g["success"][t])), t
    f = from_soa(env.state)
    np.testing.assert_array_equal(f[0, :35].view(np.uint32), g["final_state"][:35].view(np.uint32))


def test_label_transitions_golden(hx, golden_dir):
    g = np.load(os.path.join(golden_dir, "env_getreward.npz"))
    n = g["s"].shape[0]
    r = torch.zeros(n, device="cuda")
    sc = torch.zeros(n, dtype=torch.int8, device="cuda")
    dn = torch.zeros(n, dtype=torch.uint8, device="cuda")
    s, a, ns = (torch.from_numpy(g[k]).cuda() for k in ("s", "a", "ns"))
    hx.lib.call("hx_label_transitions", s.data_ptr(), a.data_ptr(), ns.data_ptr(), n, r.data_ptr(), sc.data_ptr(), dn.data_ptr(),
                hx.lib.stream_ptr())
    assert_ref_close(r.cpu().numpy(), g["reward"], "label reward")
    np.testing.assert_array_equal(sc.cpu().numpy(), g["success"])
    np.testing.assert_array_equal(dn.cpu().numpy(), g["done"])


def test_wrapper_edge_states_match_reference_rules(hx):
    """States built to sit exactly on the wrapper's thresholds: altitude 500/2000/7000/10000 +- 1 ulp-ish, health at
    0.1f vs just below, fire with/without slot and lock.  Checked against the oracle (whose wrapper layer is
    pinned to the reference on the same edge cases, tests/test_oracle_env.py)."""
    rng = np.random.default_rng(3)
    n = 4096
    e = random_states(n, rng, np.zeros(n, np.int64))
    e[:, 3:6] = 0.0  # no motion: the altitude after the tick stays where we put it, up to gravity's v*dt
    e[:, 4] = 9.8 / 60  # cancel gravity's first velocity increment exactly enough to sit near the edge
    edges = np.array([500, 2000, 7000, 10000], np.float32)
    e[:, 1] = np.repeat(edges, n // 4) + rng.choice(np.array([-0.01, -0.001, 0.0, 0.001, 0.01], np.float32), n)
    a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    env = hx.Env(n, scenario=0, auto_reset=False)
    env.set_state(to_soa(e))
    obs, r, d, s = env.step(torch.from_numpy(a).cuda())
    eo, oo = e.copy(), np.zeros((n, 13), np.float32)
    ro, do, so = ox.step_batch(eo, a, oo)
    np.testing.assert_array_equal(d.cpu().numpy(), do)
    np.testing.assert_array_equal(s.cpu().numpy(), so)
    assert_float_close(r.cpu().numpy(), ro, "reward")
    assert 0 < do.sum() < n


def test_full_size_mixed_config_properties(hx):
    """BASELINE.json configs[4] shape on one GPU: 131,072 envs, scenario = id mod 3 sorted into contiguous thirds, 25 steps with
    auto-reset and fused insert.  Size-independent properties + exact oracle parity on a 3,000-env sample:
      - a launch over the whole set == the same envs stepped as two half-size shards (no cross-env coupling, shard-safe);
      - same seed -> bit-identical state (determinism);
      - attitude quaternions stay unit length, counters advance by one per step, stats add up to envs x steps;
      - every stored replay row is exactly (prev obs, action, obs, reward, done) of some env."""
    n, steps, max_step = 131072, 25, 10
    scen = np.sort(np.arange(n) % 3).astype(np.int32)
    rng = np.random.default_rng(5)
    acts = [torch.from_numpy(rng.uniform(-1, 1, (n, 4)).astype(np.float32)).cuda() for _ in range(steps)]
    sample = np.sort(rng.choice(n, 3000, replace=False))

    def run(lo, hi, with_replay):
        rep = hx.Replay(n * steps) if with_replay else None
        env = hx.Env(hi - lo, scenario=scen[lo:hi], seed=11, max_step=max_step, auto_reset=True, random_reset=True, env_id0=lo, replay=rep)
        env.reset()
        obs_hist = [env.obs.clone()]
        outs = []
        for t in range(steps):
            o, r, d, s = env.step(acts[t][lo:hi].contiguous())
            obs_hist.append(o.clone())
            outs.append((r.clone(), d.clone(), s.clone()))
        torch.cuda.synchronize()
        return env, rep, obs_hist, outs

    env, rep, obs_hist, outs = run(0, n, True)
    # two shards == whole
    half = n // 2
    ea, _, _, _ = run(0, half, False)
    eb, _, _, _ = run(half, n, False)
    assert torch.equal(env.state[:, :half], ea.state) and torch.equal(env.state[:, half:], eb.state)
    # determinism
    env2, _, _, _ = run(0, n, False)
    assert torch.equal(env.state, env2.state)
    st = env.state
    for q0 in (6, 19):
        norm = (st[q0:q0 + 4] ** 2).sum(0).sqrt()
        assert float((norm - 1).abs().max()) < 1e-5
    sd = env.stats_dict()
    assert sd["env_steps"] == n * steps and sd["episodes"] >= 2 * n and sd["time_limit"] <= sd["episodes"]
    stored = int(rep.total.item())
    assert stored == n * steps - sd["time_limit"]  # exactly the steps that hit the time limit are not stored (train_all.py:346-347)
    # oracle parity on the sample: replay the same 25 steps env by env
    envs, oobs = ox.reset_batch(len(sample), 0, 1, seed=11)  # placeholder shapes
    L = ox.lib()
    for j, i in enumerate(sample):
        L.ox_env_reset(envs[j].ctypes.data, int(scen[i]), 1, 11, int(i), 0)
        L.ox_env_observe(envs[j].ctypes.data, oobs[j].ctypes.data)
    epi = np.zeros(len(sample), np.uint32)
    np.testing.assert_allclose(obs_hist[0].cpu().numpy()[sample], oobs, rtol=RTOL, atol=ATOL)
    # the oracle batch call numbers envs consecutively from env_id0; step each sampled env as its own batch of 1
    for t in range(steps):
        a = acts[t].cpu().numpy()[sample]
        r_g, d_g, s_g = (x.cpu().numpy()[sample] for x in outs[t])
        o_g = obs_hist[t + 1].cpu().numpy()[sample]
        for j, i in enumerate(sample):
            ro, do, so = ox.step_batch(envs[j:j + 1], a[j:j + 1], oobs[j:j + 1], max_step=max_step, auto_reset=1, randomize=1, seed=11,
                                       env_id0=int(i), episode_ctr=epi[j:j + 1])
            assert (int(do[0]), int(so[0])) == (int(d_g[j]), int(s_g[j])), (t, i)
            assert abs(ro[0] - r_g[j]) <= RTOL * abs(ro[0]) + ATOL
        np.testing.assert_allclose(o_g, oobs, rtol=RTOL, atol=ATOL, err_msg=f"step {t}")
    np.testing.assert_array_equal(from_soa(env.state)[sample].view(np.uint32), envs.view(np.uint32))
    # every stored row is one env's transition: its 'next obs' columns appear in that step's observations (spot check 2,000 rows)
    rows = rep.ring[:stored].cpu().numpy()
    pick = rng.choice(stored, 2000, replace=False)
    assert np.all((rows[pick, 31] == 0) | (rows[pick, 31] == 1)) and np.all(np.abs(rows[pick, 13:17]) <= 1)
    assert np.all(np.isfinite(rows[pick]))


def test_nonfinite_actions_are_neutralised(hx):
    """A NaN / Inf action component (a diverged policy) is taken as 0 before it reaches the state, counted in the statistics and
    stored as 0 in the replay row — GPU and oracle agree bit for bit, and nothing non-finite appears anywhere."""
    n = 5000
    rng = np.random.default_rng(11)
    scen = rng.integers(0, 3, n)
    e = random_states(n, rng, scen)
    a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    bad = rng.random((n, 4)) < 0.05
    a[bad] = rng.choice(np.array([np.nan, np.inf, -np.inf], np.float32), int(bad.sum()))
    obs_prev = rng.uniform(-1, 1, (n, 13)).astype(np.float32)
    rep = hx.Replay(1 << 14, "cuda")
    env = hx.Env(n, scenario=0, auto_reset=False, max_step=0, collect_stats=True, replay=rep)
    env.set_state(to_soa(e), torch.from_numpy(obs_prev))
    obs, r, d, s = env.step(torch.from_numpy(a).cuda())
    torch.cuda.synchronize()
    eo, oo = e.copy(), obs_prev.copy()
    stats = np.zeros(9, np.uint64)
    ring, total = np.zeros((1 << 14, 32), np.float32), np.zeros(1, np.uint64)
    ro, do, so = ox.step_batch(eo, a, oo, stats=stats, ring=ring, total=total)
    g = from_soa(env.state)
    assert np.isfinite(g[:, :35]).all() and np.isfinite(obs.cpu().numpy()).all() and np.isfinite(r.cpu().numpy()).all()
    np.testing.assert_array_equal(g.view(np.uint32), eo.view(np.uint32))
    assert_float_close(obs.cpu().numpy(), oo, "obs")
    assert_float_close(r.cpu().numpy(), ro, "reward")
    np.testing.assert_array_equal(d.cpu().numpy(), do)
    st = env.stats_dict()
    assert st["nonfinite_actions"] == int(stats[8]) == int(bad.any(1).sum()) > 100
    rows = rep.ring[:n].cpu().numpy()
    assert np.isfinite(rows).all()
    key = lambda x: np.lexsort(x.view(np.uint32).T[::-1])  # noqa: E731
    np.testing.assert_array_equal(rows[key(rows)].view(np.uint32), ring[:n][key(ring[:n])].view(np.uint32))


def test_state_pitch_is_only_an_address_matter(hx):
    """The struct-of-arrays state may sit at any pitch >= n (the C ABI's `stride`; from 262,144 envs on the default adds 1,056 floats,
    tools/ubench/env_pitch.py): same bits as the dense layout, for the stand-alone env kernel (both launch shapes) and through a snapshot."""
    n, steps = 5000, 12
    rng = np.random.default_rng(9)
    acts = [torch.from_numpy(rng.uniform(-1, 1, (n, 4)).astype(np.float32)).cuda() for _ in range(steps)]

    def run(pitch, layout):
        rep = hx.Replay(n * steps)
        env = hx.Env(n, scenario="serpentine", seed=3, max_step=7, auto_reset=True, random_reset=True, replay=rep, pitch=pitch, layout=layout)
        env.reset()
        outs = []
        for a in acts:
            o, r, d, s = env.step(a)
            outs.append((o.clone(), r.clone(), d.clone(), s.clone()))
        torch.cuda.synchronize()
        return env, rep, outs

    from hirl4ucav_amd import _lib
    e0, r0, o0 = run(0, 0)
    assert e0.pitch == n and hx.Env(1 << 18, scenario="circular").pitch == (1 << 18) + 1056
    for pitch, layout in ((n + 1056, 0), (n + 32, _lib.layout(True, 64)), (2 * n, _lib.layout(False, 256))):
        e1, r1, o1 = run(pitch, layout)
        assert e1.pitch == pitch and e1.state.shape == e0.state.shape
        assert torch.equal(e1.state, e0.state)
        rows = [r.ring.cpu().numpy().view(np.uint32) for r in (r0, r1)]  # (workgroups reserve their ring slots in any order: same rows, as a set)
        rows = [a[np.lexsort(a.T[::-1])] for a in rows]
        np.testing.assert_array_equal(rows[0], rows[1])
        for a, b in zip(o0, o1):
            assert all(torch.equal(x, y) for x, y in zip(a, b))
        assert float(e1._state_store[:, n:].abs().max()) == 0.0  # nothing written into the padding
