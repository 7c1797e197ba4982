"""CPU tests that PIN the env oracle (oracle/env_oracle.c) against golden vectors produced by the reference
wrapper itself (tests/golden/gen_env_golden.py; reference hirl/environments/HarfangEnv_GYM.py).

Tolerance: the reference computes obs/reward in float64 from the simulator's numbers; the oracle and the
GPU path are fp32 end to end.  Floats: rtol 1e-5 (the north_star's bound), atol 1e-6.  Masks: exact.
"""
import os

import numpy as np
import pytest

from tests import _oracle as ox

RTOL, ATOL = 1e-5, 1e-6
SCEN = {"HarfangEnv": 0, "HarfangSerpentineEnv": 1, "HarfangCircularEnv": 2, "HarfangSerpentineInfiniteEnv": 1}


def test_philox_known_answers():
    """Philox4x32-10 known-answer vectors (Random123 kat_vectors)."""
    L = ox.lib()

    def run(ctr, key):
        c, k, o = np.asarray(ctr, np.uint32), np.asarray(key, np.uint32), np.zeros(4, np.uint32)
        L.ox_philox4x32_10(ox.p(c), ox.p(k), ox.p(o))
        return [int(x) for x in o]

    assert run([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert run([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == [
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    r = run([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2)
    assert (r[0], r[1], r[3]) == (0x408F276D, 0x41C83B0E, 0x6D5451FD)


@pytest.mark.parametrize("cls", list(SCEN))
def test_wrapper_layer_matches_reference_trace(cls, golden_dir):
    """E4, E6, E7, E8, E9, E10-E12: replay the scripted read-backs through the oracle's wrapper layer."""
    g = np.load(os.path.join(golden_dir, f"env_wrapper_{cls}.npz"))
    L = ox.lib()
    flags = np.zeros(1, np.uint32)
    counters = np.zeros(1, np.uint32)
    obs = np.zeros(13, np.float32)
    succ = np.zeros(1, np.int8)
    cmd = np.zeros(3, np.float32)
    n = g["obs"].shape[0]
    tick = 0
    inf_fire = inf_succ = 0
    n_fire_rows = 0
    for k in range(n):
        rb = ox.make_readback(g["readback"][k])
        if g["is_reset"][k]:
            # reset(): latches cleared, then reset's own _get_observation  HarfangEnv_GYM.py:34-49
            flags[0] = ox.F_SLOT_PREV | ox.F_SLOT | (SCEN[cls] << ox.F_SCEN_SHIFT)
            counters[0] = 0
            L.ox_wrap_observe(ox.p(flags), ox.p(rb), ox.p(obs))
            np.testing.assert_allclose(obs, g["obs"][k], rtol=RTOL, atol=ATOL)
            assert not (flags[0] & (ox.F_DONE | ox.F_EPISODE_SUCCESS | ox.F_FIRE_SUCCESS))
            continue
        a = g["actions"][k]
        L.ox_script_opponent(ox.p(flags), ox.p(counters), ox.p(cmd))
        np.testing.assert_array_equal(cmd, g["opp_cmd"][tick])
        fire = a[3] > 0
        assert bool(fire) == bool(g["fired"][tick])
        tick += 1
        flags[0] = (flags[0] | ox.F_FIRED) if fire else (flags[0] & ~np.uint32(ox.F_FIRED))
        L.ox_wrap_observe(ox.p(flags), ox.p(rb), ox.p(obs))
        r = L.ox_wrap_reward(ox.p(flags), ox.p(rb), ox.p(succ))
        L.ox_wrap_terminate(ox.p(flags), ox.p(rb))
        f = int(flags[0])
        np.testing.assert_allclose(obs, g["obs"][k], rtol=RTOL, atol=ATOL, err_msg=f"obs row {k}")
        np.testing.assert_allclose(r, g["reward"][k], rtol=RTOL, atol=ATOL, err_msg=f"reward row {k}")
        assert bool(f & ox.F_DONE) == bool(g["done"][k]), k
        assert int(succ[0]) == int(g["success"][k]), k
        assert bool(f & ox.F_FIRED) == bool(g["now_missile"][k])
        assert bool(f & ox.F_SLOT_PREV) == bool(g["missile1"][k])
        assert bool(f & ox.F_SLOT) == bool(g["n_missile1"][k])
        assert bool(f & ox.F_LOCKED_PREV) == bool(g["locked_prev"][k])
        assert bool(f & ox.F_EPISODE_SUCCESS) == bool(g["episode_success"][k]), k
        assert bool(f & ox.F_FIRE_SUCCESS) == bool(g["fire_success"][k]), k
        inf_fire += int(succ[0] != 0)
        inf_succ += int(succ[0] == 1)
        n_fire_rows += int(succ[0] != 0)
        if cls == "HarfangSerpentineInfiniteEnv":
            assert (inf_fire, inf_succ) == (int(g["inf_fire"][k]), int(g["inf_success"][k]))
    assert n_fire_rows >= 5 and g["done"].sum() > 0  # the trace really exercises fire + termination paths
    if cls == "HarfangSerpentineInfiniteEnv":
        # rearm before every 60th step_test  HarfangEnv_GYM.py:484-486
        np.testing.assert_array_equal(g["rearm_ticks"][1:], np.arange(59, tick, 60))


def test_opponent_command_streams(golden_dir):
    """E10 / E11 over 2,000 steps: serpentine flips at call 250 then every 500; circular pitch switch at 100."""
    g = np.load(os.path.join(golden_dir, "env_opponent_stream.npz"))
    L = ox.lib()
    for name, scen in (("straight_line", 0), ("serpentine", 1), ("circular", 2)):
        flags = np.array([scen << ox.F_SCEN_SHIFT], np.uint32)
        counters = np.zeros(1, np.uint32)
        cmd = np.zeros(3, np.float32)
        out = []
        for _ in range(g[name].shape[0]):
            L.ox_script_opponent(ox.p(flags), ox.p(counters), ox.p(cmd))
            out.append(cmd.copy())
        np.testing.assert_array_equal(np.asarray(out), g[name])
    s = g["serpentine"][:, 2]
    assert s[248] == np.float32(-0.1) and s[249] == np.float32(0.1) and s[748] == np.float32(0.1) and s[749] == np.float32(-0.1)
    c = g["circular"]
    assert c[98, 0] == np.float32(-0.02) and c[99, 0] == np.float32(-0.01) and np.all(c[:, 1] == np.float32(0.28))


@pytest.mark.parametrize("tag", ["straight_line", "serpentine", "circular"])
def test_closed_loop_matches_reference_over_oracle_sim(tag, golden_dir):
    """The reference wrapper stepped over the oracle SIMULATOR (fake dogfight_client) vs the oracle's own fused
    step (ox_env_step = E4..E9) replayed from the recorded actions."""
    g = np.load(os.path.join(golden_dir, f"env_closedloop_{tag}.npz"))
    L = ox.lib()
    env = np.zeros(37, np.float32)
    obs = np.zeros(13, np.float32)
    L.ox_env_reset(ox.p(env), int(g["scenario"]), 0, 0, 0, 0)
    L.ox_env_observe(ox.p(env), ox.p(obs))
    np.testing.assert_allclose(obs, g["obs0"], rtol=RTOL, atol=ATOL)
    r = np.zeros(1, np.float32)
    d = np.zeros(1, np.uint8)
    s = np.zeros(1, np.int8)
    for t in range(g["actions"].shape[0]):
        a = g["actions"][t].copy()
        L.ox_env_step(ox.p(env), ox.p(a), ox.p(obs), ox.p(r), ox.p(d), ox.p(s))
        np.testing.assert_allclose(obs, g["obs"][t], rtol=RTOL, atol=ATOL, err_msg=f"step {t}")
        np.testing.assert_allclose(r[0], g["reward"][t], rtol=RTOL, atol=ATOL, err_msg=f"step {t}")
        assert (int(d[0]), int(s[0])) == (int(g["done"][t]), int(g["success"][t])), t
        f = int(env.view(np.uint32)[35])
        assert bool(f & ox.F_EPISODE_SUCCESS) == bool(g["episode_success"][t])
        assert bool(f & ox.F_FIRE_SUCCESS) == bool(g["fire_success"][t])
    # the simulator state words (all but the wrapper's flags/counters) are bit-identical
    np.testing.assert_array_equal(env[:35].view(np.uint32), g["final_state"][:35].view(np.uint32))
    if tag != "serpentine":
        assert g["done"][-1] and g["episode_success"][-1] and g["reward"][-1] > 500  # lock -> fire -> kill -> +600


def test_get_reward_and_termination(golden_dir):
    """E13: expert-labelling reward/termination  HarfangEnv_GYM.py:299-336."""
    g = np.load(os.path.join(golden_dir, "env_getreward.npz"))
    L = ox.lib()
    sc = np.zeros(1, np.int8)
    for i in range(g["s"].shape[0]):
        s, a, ns = g["s"][i].copy(), g["a"][i].copy(), g["ns"][i].copy()
        r = L.ox_get_reward(ox.p(s), ox.p(a), ox.p(ns), ox.p(sc))
        np.testing.assert_allclose(r, g["reward"][i], rtol=RTOL, atol=ATOL)
        assert int(sc[0]) == int(g["success"][i])
        assert bool(L.ox_get_termination(ox.p(ns))) == bool(g["done"][i])


def test_random_reset_distribution(golden_dir):
    """E3: random.randint(-100, 100) per axis -> integer offsets, inclusive bounds (G3).  The build draws them
    from Philox; same support and integer-valued, not the same stream."""
    g = np.load(os.path.join(golden_dir, "env_random_reset.npz"))
    ref_off = g["ally_xyz"] - np.array([0, 3500, -4000.0])
    assert ref_off.min() == -100 and ref_off.max() == 100 and np.all(ref_off == np.round(ref_off))
    envs, obs = ox.reset_batch(4000, 0, 1, seed=123)
    off = envs[:, 0:3].astype(np.float64) - np.array([0, 3500, -4000.0])
    assert off.min() == -100 and off.max() == 100 and np.all(off == np.round(off))
    assert abs(off.mean()) < 3 and abs(off.std() - np.sqrt((201 ** 2 - 1) / 12)) < 2
    # opponent pose / speeds / health after reset  HarfangEnv_GYM.py:68-81
    assert np.all(envs[:, 13:16] == np.array([0, 4200, 0], np.float32)) and np.all(envs[:, 32] == np.float32(0.2))
    assert np.all(envs[:, 3:6] == np.array([0, 0, 300], np.float32)) and np.all(envs[:, 16:19] == np.array([0, 0, 200], np.float32))
    e2, _ = ox.reset_batch(2, 2, 0, seed=0)
    assert np.all(e2[:, 16:19] == np.array([0, 0, 290], np.float32))  # circular: 290 m/s  :472-473
    # different (env, episode) -> different draws; same -> same
    a, _ = ox.reset_batch(8, 0, 1, seed=9, env_id0=5, episode=3)
    b, _ = ox.reset_batch(8, 0, 1, seed=9, env_id0=5, episode=3)
    c, _ = ox.reset_batch(8, 0, 1, seed=9, env_id0=5, episode=4)
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_batch_step_episode_rules():
    """Vectorised-driver rules restated from train_all.py:341-361: the step that reaches max_step is executed but
    not stored, the episode then restarts; done transitions ARE stored."""
    n, cap, max_step = 6, 64, 5
    envs, obs = ox.reset_batch(n, [0, 1, 2, 0, 1, 2], 1, seed=1)
    ring = np.zeros((cap, 32), np.float32)
    rsucc = np.zeros(cap, np.int8)
    total = np.zeros(1, np.uint64)
    stats = np.zeros(9, np.uint64)
    epi = np.zeros(n, np.uint32)
    rng = np.random.default_rng(0)
    stored = 0
    for t in range(12):
        a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
        prev = obs.copy()
        r, d, s = ox.step_batch(envs, a, obs, max_step=max_step, auto_reset=1, randomize=1, seed=1, episode_ctr=epi,
                                ring=ring, ring_succ=rsucc, total=total, stats=stats)
        trunc = (t + 1) % max_step == 0
        if not trunc:
            rows = ring[stored:stored + n]
            np.testing.assert_array_equal(rows[:, :13], prev)
            np.testing.assert_array_equal(rows[:, 13:17], a)
            np.testing.assert_array_equal(rows[:, 17:30], obs)
            np.testing.assert_array_equal(rows[:, 30], r)
            stored += n
        assert int(total[0]) == stored
    assert int(stats[0]) == 2 * n and int(stats[3]) == 2 * n and np.all(epi == 2)


def test_model_trigonometry_accuracy_and_special_values():
    """docs/DYNAMICS.md "Angles": the read-back's asin / acos / atan2 are the model's own polynomials (so that host and gfx950 agree bit
    for bit); against float64 they stay within 4e-7 rad, and the quadrant / end-point values are exact enough for the wrapper."""
    L = ox.lib()
    xs = np.linspace(-1, 1, 40001).astype(np.float32)
    assert max(abs(L.ox_asin(float(x)) - np.arcsin(np.float64(x))) for x in xs) < 4e-7
    assert max(abs(L.ox_acos(float(x)) - np.arccos(np.float64(x))) for x in xs) < 4e-7
    rng = np.random.default_rng(0)
    pts = np.concatenate([rng.normal(size=(20000, 2)), rng.normal(size=(2000, 2)) * [1e-3, 1.0], rng.normal(size=(2000, 2)) * [1.0, 1e-3]]).astype(np.float32)
    assert max(abs(L.ox_atan2(float(y), float(x)) - np.arctan2(np.float64(y), np.float64(x))) for y, x in pts) < 4e-7
    assert L.ox_atan2(0.0, 1.0) == 0.0 and L.ox_atan2(0.0, 0.0) == 0.0 and L.ox_asin(0.0) == 0.0 and L.ox_acos(1.0) == 0.0
    assert abs(L.ox_atan2(1.0, 0.0) - np.pi / 2) < 2e-7 and abs(L.ox_atan2(0.0, -1.0) - np.pi) < 3e-7 and abs(L.ox_acos(-1.0) - np.pi) < 3e-7
    assert L.ox_atan2(-1.0, 0.0) == -L.ox_atan2(1.0, 0.0) and L.ox_asin(-0.3) == -L.ox_asin(0.3)
