"""Physical sanity of the re-derived simulator tick (E5, docs/DYNAMICS.md "Sanity bands").

Dynamics parity with the external Harfang sandbox is UNPINNED (its source is not in the reference), and the GPU-vs-oracle tests
only prove that two implementations of docs/DYNAMICS.md agree.  These tests pin what the MODEL itself must do, on the CPU oracle
and — the same scenarios, the same bands — on the HIP kernels: trimmed flight holds altitude and speed, full-stick turn rates
follow the documented rate constants, a guided missile launched inside the envelope kills within its lifetime, an unguided one
flies straight and misses."""
import numpy as np
import pytest

from tests import _oracle as ox

DT = 1.0 / 60.0


class OracleSim:
    """n envs stepped by the CPU oracle (AoS state [n, 37])"""

    def __init__(self, scen, n=1):
        self.envs, self.obs = ox.reset_batch(n, scen, 0, seed=0)

    def state(self):
        return self.envs

    def set_state(self, e):
        self.envs[:] = e

    def step(self, a):
        r, d, s = ox.step_batch(self.envs, np.asarray(a, np.float32), self.obs)
        return self.obs, r, d, s


class GpuSim:
    """the same interface over hx_env_step (state transposed to the oracle's layout for the tests to read)"""

    def __init__(self, scen, n=1):
        import torch

        from hirl4ucav_amd.environments.batched import BatchedHarfangEnv

        self.torch = torch
        scen = np.broadcast_to(np.asarray(scen, np.int32), (n,)).copy()
        self.env = BatchedHarfangEnv(n, scenario=scen, seed=0, auto_reset=False, random_reset=False, collect_stats=False)
        self.obs = self.env.reset().cpu().numpy()

    def state(self):
        return np.ascontiguousarray(self.env.state.cpu().numpy().T)

    def set_state(self, e):
        self.env.set_state(self.torch.from_numpy(np.ascontiguousarray(e.T)))

    def step(self, a):
        o, r, d, s = self.env.step(self.torch.from_numpy(np.asarray(a, np.float32)).cuda())
        self.obs = o.cpu().numpy()
        return self.obs, r.cpu().numpy(), d.cpu().numpy(), s.cpu().numpy()


BACKENDS = [pytest.param(OracleSim, id="oracle"), pytest.param(GpuSim, id="gpu", marks=pytest.mark.gpu)]


def make(backend, scen, n=1):
    if backend is GpuSim:
        torch = pytest.importorskip("torch")
        if not torch.cuda.is_available():
            pytest.skip("no GPU")
    return backend(scen, n)


def speed(e, w0):
    return np.linalg.norm(e[:, w0 + 3:w0 + 6], axis=1)


# ---- 1. trimmed, hands-off flight -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("backend", BACKENDS)
def test_hands_off_flight_stays_in_its_bands(backend):
    """No stick input for a whole episode (1,500 / 1,900 steps) in the three scenarios at once: the ally (thrust 1.0, 300 m/s, 3,500 m)
    holds 3,500 +- 80 m and 290..305 m/s; the straight / serpentine opponent (thrust 0.6, 200 m/s, 4,200 m) holds 4,200 +- 80 m and
    settles between 195 and 245 m/s; the circling opponent (thrust 0.8, 290 m/s, roll 0.28, pitch -0.02 / -0.01) climbs inside
    4,150..4,900 m at 250..295 m/s.  Nobody leaves the 500..10,000 m band (no done), attitudes stay unit quaternions."""
    sim = make(backend, [0, 1, 2], 3)
    a = np.tile(np.array([0, 0, 0, -1], np.float32), (3, 1))
    lo = np.full((3, 4), np.inf)
    hi = np.full((3, 4), -np.inf)
    for t in range(1900):
        obs, r, d, s = sim.step(a)
        if t % 10 == 0 or t > 1880:
            e = sim.state()
            live = np.array([t < 1500, t < 1500, True])
            v = np.stack([e[:, 1], speed(e, 0), e[:, 14], speed(e, 13)], 1)
            lo[live] = np.minimum(lo, v)[live]
            hi[live] = np.maximum(hi, v)[live]
            assert not d[live].any(), t
            q = np.stack([np.linalg.norm(e[:, 6:10], axis=1), np.linalg.norm(e[:, 19:23], axis=1)])
            assert np.abs(q - 1).max() < 1e-5
    assert (lo[:, 0] >= 3420).all() and (hi[:, 0] <= 3580).all(), (lo[:, 0], hi[:, 0])      # ally altitude
    assert (lo[:, 1] >= 290).all() and (hi[:, 1] <= 305).all(), (lo[:, 1], hi[:, 1])         # ally speed
    assert (lo[:2, 2] >= 4120).all() and (hi[:2, 2] <= 4280).all(), (lo[:2, 2], hi[:2, 2])   # straight / serpentine opponent altitude
    assert (lo[:2, 3] >= 195).all() and (hi[:2, 3] <= 245).all(), (lo[:2, 3], hi[:2, 3])
    assert lo[2, 2] >= 4150 and hi[2, 2] <= 4900 and lo[2, 3] >= 250 and hi[2, 3] <= 295, (lo[2], hi[2])


# ---- 2. full-stick rates ----------------------------------------------------------------------------------------------------------
def control_effectiveness(alt, spd):
    """docs/DYNAMICS.md step 2: eff = q / (q + 4000), q = 0.6125 sigma s^2, sigma = x^4 (3 + x) / 4, x = 1 - 2.2558e-5 alt"""
    x = 1 - 2.2558e-5 * alt
    q = 0.6125 * x ** 4 * (0.75 + 0.25 * x) * spd ** 2
    return q / (q + 4000.0)


@pytest.mark.parametrize("backend", BACKENDS)
def test_full_stick_turn_rates_follow_the_rate_constants(backend):
    """Full pitch / yaw / roll stick from level flight: the level slews in at 3 /s (0.05 per tick), then the body rates are
    0.8 / 0.4 / 2.5 rad/s times the control effectiveness (docs/DYNAMICS.md steps 1-3): pitch and heading advance by the documented
    rate within 10 % between t = 0.5 s and t = 1 s; the roll angle — opposed by the wing leveller — passes 1.3..1.6 rad at t = 1 s."""
    sim = make(backend, 0, 3)
    a = np.array([[1, 0, 0, -1], [0, 0, 1, -1], [0, 1, 0, -1]], np.float32)  # env 0 pitch, 1 yaw, 2 roll
    ang = []
    for t in range(60):
        obs, *_ = sim.step(a)
        ang.append(obs[:, 3:6] * np.pi)
        if t == 19:
            lv = sim.state()
            assert abs(lv[0, 10] - 1.0) < 1e-6 and abs(lv[1, 12] - 1.0) < 1e-6 and abs(lv[2, 11] - 1.0) < 1e-6  # 20 ticks x 0.05
    ang = np.array(ang)
    e = sim.state()
    eff = control_effectiveness(e[0, 1], speed(e, 0)[0])
    pitch_rate = (ang[59, 0, 0] - ang[29, 0, 0]) / (30 * DT)
    yaw_rate = (ang[59, 1, 1] - ang[29, 1, 1]) / (30 * DT)
    assert abs(pitch_rate - 0.8 * eff) < 0.1 * 0.8 * eff, (pitch_rate, 0.8 * eff)
    assert abs(yaw_rate - 0.4 * control_effectiveness(e[1, 1], speed(e, 0)[1])) < 0.04, yaw_rate
    assert 1.3 < ang[59, 2, 2] < 1.6, ang[59, 2, 2]
    assert np.abs(ang[59, 0, 1:]).max() < 1e-6 and np.abs(ang[59, 1, [0, 2]]).max() < 1e-6  # pure pitch / pure yaw stay pure


# ---- 3. missile ----------------------------------------------------------------------------------------------------------------------
def tail_chase(sim, lateral):
    """ally 1,500 m behind the opponent at its altitude, nose along +Z; the opponent `lateral` metres to the side"""
    e = sim.state().copy()
    e[:, 0:3] = (0.0, 4200.0, -1500.0)
    e[:, 13:16] = (lateral, 4200.0, 0.0)
    sim.set_state(e)


@pytest.mark.parametrize("backend", BACKENDS)
def test_guided_missile_kills_and_unguided_one_misses(backend):
    """Opponent 300 m off the nose line at 1,500 m (11 deg: inside the 15 deg cone, inside 100..3,000 m).
    env 0: hold fire until the targeting device has been in-cone for 1 s (lock), launch: the missile is GUIDED, turns onto the
    opponent and kills it (health 0.2 -> 0, done, +600) well inside its 20 s life; the wrapper reports success = +1 at the launch.
    env 1: launch on the first tick, before any lock: success = -1, the missile is UNGUIDED, keeps its launch heading and passes the
    opponent at more than the 40 m hit radius; health stays 0.2 and the missile is removed after 20 s."""
    sim = make(backend, 0, 2)
    tail_chase(sim, 300.0)
    fired = [False, False]
    succ_at_launch = [0, 0]
    min_dist = np.inf
    kill_step = None
    for t in range(1500):
        e = sim.state()
        a = np.tile(np.array([0, 0, 0, -1], np.float32), (2, 1))
        if not fired[0] and sim.obs[0, 7] > 0:  # locked (the observation the pilot sees before acting)
            a[0, 3] = 1.0
        if not fired[1] and t == 0:
            a[1, 3] = 1.0
        _, r, d, s = sim.step(a)
        for i in (0, 1):
            if a[i, 3] > 0:
                fired[i], succ_at_launch[i] = True, int(s[i])
        e = sim.state()
        flags = e.view(np.uint32)[:, 35]
        if fired[1] and (flags[1] & ox.F_M_ACTIVE):
            min_dist = min(min_dist, float(np.linalg.norm(e[1, 26:29] - e[1, 13:16])))
        if d[0] and kill_step is None:
            kill_step = t
            assert r[0] > 500 and e[0, 32] == 0.0 and (flags[0] & ox.F_EPISODE_SUCCESS)
        if kill_step is not None and t > kill_step + 5 and not (flags[1] & ox.F_M_ACTIVE) and t > 1300:
            break
    assert succ_at_launch == [1, -1]
    assert kill_step is not None and 60 <= kill_step <= 60 + int(20 / DT), kill_step   # lock delay 1 s, then at most the missile's life
    e = sim.state()
    flags = e.view(np.uint32)[:, 35]
    assert flags[0] & ox.F_M_GUIDED and not (flags[1] & ox.F_M_GUIDED)
    assert e[1, 32] == np.float32(0.2) and min_dist > 40.0, (e[1, 32], min_dist)
    assert not (flags[1] & ox.F_M_ACTIVE) and e[1, 34] > 20.0   # removed at the end of its life


@pytest.mark.parametrize("backend", BACKENDS)
def test_lock_needs_cone_range_and_one_second(backend):
    """Targeting device (docs/DYNAMICS.md): in-cone iff angle < 15 deg and 100 m < range < 3,000 m; lock after 1 s in-cone; it drops the
    tick the target leaves the cone.  Three geometries: on the nose at 1,500 m (locks at tick 60), 20 deg off (never), on the nose at
    3,500 m (never while out of range)."""
    sim = make(backend, 0, 3)
    e = sim.state().copy()
    e[:, 0:3] = (0.0, 4200.0, -1500.0)
    e[0, 13:16] = (0.0, 4200.0, 0.0)
    e[1, 13:16] = (1500.0 * np.tan(np.radians(20.0)), 4200.0, 0.0)
    e[2, 13:16] = (0.0, 4200.0, 2000.0)
    e[:, 16:19] = (0.0, 0.0, 300.0)  # the opponents keep the ally's speed: the geometry holds
    sim.set_state(e)
    a = np.tile(np.array([0, 0, 0, -1], np.float32), (3, 1))
    first_lock = [None, None, None]
    for t in range(200):
        obs, *_ = sim.step(a)
        for i in range(3):
            if obs[i, 7] > 0 and first_lock[i] is None:
                first_lock[i] = t
    assert first_lock[0] in (59, 60) and first_lock[1] is None and first_lock[2] is None, first_lock
    assert abs(obs[1, 6] * 180 - 20.0) < 1.0  # target_angle / 180 carries the 20 deg
