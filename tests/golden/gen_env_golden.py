#!/usr/bin/env python3
"""Generate the env golden vectors by running the REFERENCE wrapper itself.

Runs only in the development container (needs /root/reference, which never travels to the GPU box); the
.npz files it writes next to this script are the committed fixtures.

    python tests/golden/gen_env_golden.py

What is produced
  env_wrapper_<class>.npz   G1  reference HarfangEnv / Serpentine / Circular / SerpentineInfinite
                                (hirl/environments/HarfangEnv_GYM.py) stepped over a fake dogfight_client
                                whose read-backs are SCRIPTED (random walk + forced edge cases: altitude
                                499/500/2000/7000/10000/10001, health 0.1f / just-below / 0 / negative,
                                fire with and without missile and lock, lock acquired on the tick of the
                                fire).  Inputs: per-step read-backs + actions.  Outputs: obs, reward, done,
                                success, the step_test 9-tuple, episode_success, fire_success, and the
                                opponent command stream the wrapper sent before each tick.
  env_closedloop_<scen>.npz G1b the same reference classes stepped over a fake dogfight_client BACKED BY THE
                                ORACLE SIMULATOR (oracle/libhx_oracle.so ox_sim_tick / ox_sim_readback), so
                                the whole step (sim + wrapper) can be replayed from the actions alone.
  env_getreward.npz         G2  reference get_reward / get_termination on random (s, a, s').
  env_random_reset.npz      G3  reset_machine_matrix arguments of random_reset for seeds 0..9 (documents
                                the U{-100..100} integer distribution; the build uses Philox, not MT19937).
"""
import ctypes
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


# ---------------------------------------------------------------------------------------------------
# stubs for modules the reference imports but this image lacks (gym) — only Box.sample is ever used
# ---------------------------------------------------------------------------------------------------
class _Box:
    def __init__(self, low, high, dtype=np.float64):
        self.low, self.high, self.dtype = np.asarray(low), np.asarray(high), dtype

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)


gym = types.ModuleType("gym")
gym.spaces = types.ModuleType("gym.spaces")
gym.spaces.Box = _Box
sys.modules["gym"] = gym
sys.modules["gym.spaces"] = gym.spaces
sys.path.insert(0, REF)

import hirl.environments.dogfight_client as df  # noqa: E402
import hirl.environments.HarfangEnv_GYM as ref_env  # noqa: E402

DF_NAMES = [n for n in dir(df) if callable(getattr(df, n)) and not n.startswith("_")
            and getattr(getattr(df, n), "__module__", "") == df.__name__]


# ---------------------------------------------------------------------------------------------------
# fake dogfight_client: records every call; answers read-backs from a backend object
# ---------------------------------------------------------------------------------------------------
class FakeDF:
    def __init__(self, backend):
        self.b = backend
        self.log = []
        self.levels = {"ally_1": [0.0, 0.0, 0.0], "ennemy_2": [0.0, 0.0, 0.0]}  # pitch, roll, yaw
        self.fire = False

    def install(self):
        for name in DF_NAMES:
            setattr(df, name, self._make(name))

    def _make(self, name):
        def f(*args):
            self.log.append((name, args))
            h = getattr(self, "do_" + name, None)
            return h(*args) if h else None
        return f

    def do_get_machine_missiles_list(self, mid):
        return ["Meteor_0"]

    def do_set_plane_pitch(self, pid, v):
        self.levels[pid][0] = v

    def do_set_plane_roll(self, pid, v):
        self.levels[pid][1] = v

    def do_set_plane_yaw(self, pid, v):
        self.levels[pid][2] = v

    def do_fire_missile(self, pid, slot):
        self.fire = True

    def do_rearm_machine(self, pid):
        self.b.rearm()

    def do_reset_machine_matrix(self, pid, x, y, z, rx, ry, rz):
        self.b.reset_matrix(pid, x, y, z)

    def do_set_plane_linear_speed(self, pid, v):
        self.b.set_speed(pid, v)

    def do_set_plane_thrust(self, pid, v):
        self.b.set_thrust(pid, v)

    def do_set_health(self, pid, v):
        self.b.set_health(v)

    def do_update_scene(self):
        self.b.tick(self.levels["ally_1"], self.levels["ennemy_2"], self.fire)
        self.fire = False

    def do_get_plane_state(self, pid):
        return self.b.plane_state(pid)

    def do_get_health(self, pid):
        return {"health_level": self.b.health()}

    def do_get_missiles_device_slots_state(self, pid):
        return {"missiles_slots": [self.b.slot()]}


f32 = lambda x: float(np.float32(x))  # noqa: E731  every scripted value is fp32-representable


class ScriptedBackend:
    """Read-backs follow a script (random walk + forced edge cases); slot/rearm are stateful."""

    def __init__(self, seed, nsteps):
        self.rng = np.random.default_rng(seed)
        self.t = 0
        self.sl = True
        self.rb = []  # recorded read-backs, one per observation
        self.cmds = []  # opponent commands seen at each tick
        self.fires = []
        n = max(nsteps, 120) + 8
        r = self.rng
        self.ally_pos = np.cumsum(r.normal(0, 5, (n, 3)), 0) + np.array([0, 3500, -4000.0])
        self.opp_pos = np.cumsum(r.normal(0, 5, (n, 3)), 0) + np.array([0, 4200, 0.0])
        self.ally_eul = r.uniform(-1, 1, (n, 3)) * np.array([np.pi / 2, np.pi, np.pi])
        self.opp_eul = r.uniform(-1, 1, (n, 3)) * np.array([np.pi / 2, np.pi, np.pi])
        self.angle = r.uniform(0, 180, n)
        # lock flag with persistence
        lk = np.zeros(n, bool)
        state = False
        for i in range(n):
            if r.random() < 0.08:
                state = not state
            lk[i] = state
        self.locked = lk
        self.hl = np.full(n, 0.2)
        # forced edge cases on altitude (ally y) at fixed steps
        edge_alt = [499.0, 500.0, 500.5, 1999.5, 2000.0, 2000.5, 6999.5, 7000.0, 7000.5, 9999.5, 10000.0, 10001.0]
        for k, v in enumerate(edge_alt):
            self.ally_pos[40 + 3 * k, 1] = v
        # health edge cases late in the trace
        below = float(np.nextafter(np.float32(0.1), np.float32(0)))
        h_edges = [0.2, 0.1, below, 0.1, 0.15, below, 0.05, 0.0, -0.05, 0.2]
        start = max(100, nsteps - 60)
        for k, v in enumerate(h_edges):
            self.hl[start + 4 * k:start + 4 * k + 4] = v

    # wrapper-facing API --------------------------------------------------------------------
    def rearm(self):
        self.sl = True

    def reset_matrix(self, pid, x, y, z):
        pass

    def set_speed(self, pid, v):
        pass

    def set_thrust(self, pid, v):
        pass

    def set_health(self, v):
        pass

    def tick(self, ally_lv, opp_lv, fire):
        self.cmds.append(list(opp_lv))
        self.fires.append(bool(fire))
        if fire and self.sl:
            self.sl = False
        self.t += 1

    def plane_state(self, pid):
        t = self.t
        if pid == "ally_1":
            return {"position": [f32(v) for v in self.ally_pos[t]], "Euler_angles": [f32(v) for v in self.ally_eul[t]],
                    "heading": 0.0, "pitch_attitude": 0.0, "roll_attitude": 0.0,
                    "user_pitch_level": 0.0, "user_roll_level": 0.0, "user_yaw_level": 0.0,
                    "target_locked": bool(self.locked[t]), "target_angle": f32(self.angle[t])}
        return {"position": [f32(v) for v in self.opp_pos[t]], "Euler_angles": [f32(v) for v in self.opp_eul[t]],
                "heading": 0.0, "pitch_attitude": 0.0, "roll_attitude": 0.0}

    def health(self):
        return f32(self.hl[self.t])

    def slot(self):
        # called last inside _get_observation -> record the whole read-back here
        t = self.t
        self.rb.append([f32(v) for v in self.ally_pos[t]] + [f32(v) for v in self.ally_eul[t]]
                       + [f32(v) for v in self.opp_pos[t]] + [f32(v) for v in self.opp_eul[t]]
                       + [f32(self.angle[t]), f32(self.hl[t]), float(self.locked[t]), float(self.sl)])
        return self.sl


def scripted_actions(rng, n):
    a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    fire = rng.random(n) < 0.12
    a[:, 3] = np.where(fire, np.abs(a[:, 3]) + 0.01, -np.abs(a[:, 3]))
    a[::97, 3] = 0.0  # a[3] == 0 must NOT fire (strict > 0, HarfangEnv_GYM.py:150)
    return a


def run_scripted(cls_name, seed, nsteps, use_step_test, reset_at):
    be = ScriptedBackend(seed, nsteps)
    fake = FakeDF(be)
    fake.install()
    env = getattr(ref_env, cls_name)()
    rng = np.random.default_rng(seed + 1000)
    acts = scripted_actions(rng, nsteps)
    out = {k: [] for k in ("obs", "reward", "done", "success", "now_missile", "missile1", "n_missile1", "locked_prev",
                            "episode_success", "fire_success", "is_reset", "loc_diff", "inf_fire", "inf_success")}
    random.seed(seed)

    def do_reset(kind):
        o = env.reset() if kind == 0 else env.random_reset()
        out["obs"].append(np.asarray(o, np.float64)); out["reward"].append(0.0); out["done"].append(env.done)
        out["success"].append(env.success); out["now_missile"].append(env.now_missile_state)
        out["missile1"].append(env.missile1_state); out["n_missile1"].append(env.n_missile1_state)
        out["locked_prev"].append(env.Ally_target_locked); out["episode_success"].append(env.episode_success)
        out["fire_success"].append(env.fire_success); out["is_reset"].append(1 + kind); out["loc_diff"].append(0.0)
        out["inf_fire"].append(getattr(env, "infinite_total_fire", 0)); out["inf_success"].append(getattr(env, "infinite_total_success", 0))

    do_reset(0)
    used = [np.zeros(4, np.float32)]
    for t in range(nsteps):
        if t in reset_at:
            do_reset(reset_at[t])
            used.append(np.zeros(4, np.float32))
        a = acts[t]
        used.append(a)
        if use_step_test:
            o, r, d, _, now_m, m1, nm1, lk, s = env.step_test(a)
        else:
            o, r, d, _, s = env.step(a)
            now_m, m1, nm1, lk = env.now_missile_state, env.missile1_state, env.n_missile1_state, env.Ally_target_locked
        out["obs"].append(np.asarray(o, np.float64)); out["reward"].append(float(r)); out["done"].append(bool(d))
        out["success"].append(int(s)); out["now_missile"].append(bool(now_m)); out["missile1"].append(bool(m1))
        out["n_missile1"].append(bool(nm1)); out["locked_prev"].append(bool(lk))
        out["episode_success"].append(bool(env.episode_success)); out["fire_success"].append(bool(env.fire_success))
        out["is_reset"].append(0); out["loc_diff"].append(float(env.loc_diff))
        out["inf_fire"].append(getattr(env, "infinite_total_fire", 0)); out["inf_success"].append(getattr(env, "infinite_total_success", 0))
    # per tick: opponent command seen by the simulator + whether FIRE_MISSILE preceded the tick; rearm calls
    rearm_ticks = []
    tick = 0
    for name, args in fake.log:
        if name == "update_scene":
            tick += 1
        elif name == "rearm_machine":
            rearm_ticks.append(tick)
    np.savez_compressed(
        os.path.join(HERE, f"env_wrapper_{cls_name}.npz"),
        readback=np.asarray(be.rb, np.float32),  # [n_obs, 16]: ally pos3 eul3, opp pos3 eul3, angle, health, locked, slot
        actions=np.asarray(used, np.float32),  # row k is the action that produced observation k (zeros for resets)
        opp_cmd=np.asarray(be.cmds, np.float32), fired=np.asarray(be.fires, np.uint8),
        rearm_ticks=np.asarray(rearm_ticks, np.int32),
        obs=np.asarray(out["obs"], np.float64), reward=np.asarray(out["reward"], np.float64),
        done=np.asarray(out["done"], np.uint8), success=np.asarray(out["success"], np.int8),
        now_missile=np.asarray(out["now_missile"], np.uint8), missile1=np.asarray(out["missile1"], np.uint8),
        n_missile1=np.asarray(out["n_missile1"], np.uint8), locked_prev=np.asarray(out["locked_prev"], np.uint8),
        episode_success=np.asarray(out["episode_success"], np.uint8), fire_success=np.asarray(out["fire_success"], np.uint8),
        is_reset=np.asarray(out["is_reset"], np.uint8), loc_diff=np.asarray(out["loc_diff"], np.float64),
        inf_fire=np.asarray(out["inf_fire"], np.int32), inf_success=np.asarray(out["inf_success"], np.int32),
        use_step_test=np.uint8(use_step_test))
    print(cls_name, "scripted:", len(out["obs"]), "observations,", int(np.sum(np.asarray(out["success"]) != 0)), "fire events,",
          int(np.sum(out["done"])), "done rows")


def opponent_stream(cls_name, nsteps):
    """2,000-step opponent command stream (serpentine flips at 250, 750, 1250, ...; circular pitch switch at 100)."""
    be = ScriptedBackend(7, nsteps)
    fake = FakeDF(be)
    fake.install()
    env = getattr(ref_env, cls_name)()
    env.reset()
    for t in range(nsteps):
        env.step(np.array([0, 0, 0, -1.0], np.float32))
    return np.asarray(be.cmds, np.float32)


# ---------------------------------------------------------------------------------------------------
# closed loop over the oracle simulator
# ---------------------------------------------------------------------------------------------------
class OracleBackend:
    def __init__(self, scenario):
        so = os.path.join(REPO, "oracle", "libhx_oracle.so")
        self.L = ctypes.CDLL(so)
        self.L.ox_env_reset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32]
        self.L.ox_sim_tick.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        self.L.ox_sim_readback.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.L.ox_env_rearm.argtypes = [ctypes.c_void_p]
        self.scenario = scenario
        self.env = np.zeros(37, np.float32)
        self.rbuf = np.zeros(16, np.float32)
        self.pending = {}
        self.L.ox_env_reset(self.env.ctypes.data, scenario, 0, 0, 0, 0)

    def _rb(self):
        self.L.ox_sim_readback(self.env.ctypes.data, self.rbuf.ctypes.data)
        return self.rbuf

    def rearm(self):
        self.L.ox_env_rearm(self.env.ctypes.data)

    def reset_matrix(self, pid, x, y, z):
        self.pending[pid] = (x, y, z)
        if len(self.pending) == 2:
            # both machines placed -> rebuild the oracle state, then overwrite the ally position the
            # wrapper chose (random_reset draws it with Python's MT19937, which the build does not mimic)
            self.L.ox_env_reset(self.env.ctypes.data, self.scenario, 0, 0, 0, 0)
            ax, ay, az = self.pending["ally_1"]
            self.env[0:3] = (ax, ay, az)
            self.pending = {}

    def set_speed(self, pid, v):
        pass  # reset speeds are the scenario constants the oracle already applies (checked in tests)

    def set_thrust(self, pid, v):
        pass

    def set_health(self, v):
        pass

    def tick(self, ally_lv, opp_lv, fire):
        a = np.asarray(ally_lv, np.float32)
        o = np.asarray(opp_lv, np.float32)
        self.L.ox_sim_tick(self.env.ctypes.data, a.ctypes.data, o.ctypes.data, int(fire))

    def plane_state(self, pid):
        rb = self._rb()
        if pid == "ally_1":
            return {"position": [float(v) for v in rb[0:3]], "Euler_angles": [float(v) for v in rb[3:6]],
                    "heading": 0.0, "pitch_attitude": 0.0, "roll_attitude": 0.0,
                    "user_pitch_level": 0.0, "user_roll_level": 0.0, "user_yaw_level": 0.0,
                    "target_locked": bool(rb[14:15].view(np.int32)[0]), "target_angle": float(rb[12])}
        return {"position": [float(v) for v in rb[6:9]], "Euler_angles": [float(v) for v in rb[9:12]],
                "heading": 0.0, "pitch_attitude": 0.0, "roll_attitude": 0.0}

    def health(self):
        return float(self._rb()[13])

    def slot(self):
        return bool(self._rb()[15:16].view(np.int32)[0])


def pursuit_action(env_state, obs, rng, p_blind_fire):
    """A crude scripted pilot so that locks, fires and kills occur in the closed-loop trace."""
    w, x, y, z = env_state[6:10].astype(np.float64)
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    b = R.T @ (env_state[13:16] - env_state[0:3]).astype(np.float64)
    b /= np.linalg.norm(b)
    a = np.array([np.clip(-4 * b[1], -1, 1), np.clip(-2 * b[0], -1, 1), np.clip(4 * b[0], -1, 1), -1.0])
    a[:3] += rng.normal(0, 0.05, 3)
    if obs[8] > 0 and (obs[7] > 0 or rng.random() < p_blind_fire):
        a[3] = 1.0
    return np.clip(a, -1, 1).astype(np.float32)


def run_closedloop(cls_name, scenario, nsteps, tag):
    be = OracleBackend(scenario)
    fake = FakeDF(be)
    fake.install()
    env = getattr(ref_env, cls_name)()
    rng = np.random.default_rng(11 + scenario)
    o = env.reset()
    obs0 = np.asarray(o, np.float64)
    acts, obs, rew, done, succ, es, fs = [], [], [], [], [], [], []
    for t in range(nsteps):
        a = pursuit_action(be.env, o, rng, 0.002 if scenario == 1 else 0.0)
        o, r, d, _, s = env.step(a)
        acts.append(a); obs.append(np.asarray(o, np.float64)); rew.append(float(r)); done.append(bool(d)); succ.append(int(s))
        es.append(bool(env.episode_success)); fs.append(bool(env.fire_success))
        if d:
            break
    np.savez_compressed(os.path.join(HERE, f"env_closedloop_{tag}.npz"), scenario=np.int32(scenario), obs0=obs0,
                        actions=np.asarray(acts, np.float32), obs=np.asarray(obs), reward=np.asarray(rew),
                        done=np.asarray(done, np.uint8), success=np.asarray(succ, np.int8),
                        episode_success=np.asarray(es, np.uint8), fire_success=np.asarray(fs, np.uint8),
                        final_state=be.env.copy())
    print(cls_name, "closed loop:", len(rew), "steps, done =", done[-1], ", fires =", int(np.sum(np.asarray(succ) != 0)),
          ", return = %.1f" % sum(rew))


def gen_getreward(n=3000):
    be = ScriptedBackend(3, 10)
    FakeDF(be).install()
    env = ref_env.HarfangEnv()
    rng = np.random.default_rng(5)
    s = rng.uniform(-1, 1, (n, 13)).astype(np.float32)
    ns = rng.uniform(-1, 1, (n, 13)).astype(np.float32)
    a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    for arr in (s, ns):
        arr[:, 7] = np.where(rng.random(n) < 0.5, 1, -1)
        arr[:, 8] = np.where(rng.random(n) < 0.5, 1, -1)
        arr[:, 12] = rng.uniform(0, 0.2, n)
    below = np.nextafter(np.float32(0.1), np.float32(0))
    ns[:50, 12] = np.float32(0.1); ns[50:100, 12] = below; ns[100:150, 12] = 0.0
    a[::53, 3] = 0.0
    r = np.zeros(n); sc = np.zeros(n, np.int8); dn = np.zeros(n, np.uint8)
    for i in range(n):
        # the reference is fed the float64 image of the fp32 rows, as np.fromstring would give for such text
        rr, ss = env.get_reward(s[i].astype(np.float64), a[i].astype(np.float64), ns[i].astype(np.float64))
        r[i], sc[i], dn[i] = rr, ss, env.get_termination(ns[i].astype(np.float64))
    np.savez_compressed(os.path.join(HERE, "env_getreward.npz"), s=s, a=a, ns=ns, reward=r, success=sc, done=dn)
    print("get_reward:", n, "rows,", int(dn.sum()), "done,", int((sc != 0).sum()), "fire rows")


def gen_random_reset():
    rows = []
    for seed in range(10):
        be = ScriptedBackend(3, 10)
        fake = FakeDF(be)
        fake.install()
        env = ref_env.HarfangEnv()
        random.seed(seed)
        for _ in range(50):
            env.random_reset()
        rows += [args[1:4] for name, args in fake.log if name == "reset_machine_matrix" and args[0] == "ally_1"]
    rows = np.asarray(rows, np.float64)
    np.savez_compressed(os.path.join(HERE, "env_random_reset.npz"), ally_xyz=rows)
    off = rows - np.array([0, 3500, -4000.0])
    print("random_reset: offsets in [%d, %d], all integer: %s" % (off.min(), off.max(), bool(np.all(off == np.round(off)))))


if __name__ == "__main__":
    run_scripted("HarfangEnv", 1, 700, False, {100: 1, 200: 0, 300: 1, 400: 0, 500: 1, 600: 0})
    run_scripted("HarfangSerpentineEnv", 2, 700, False, {90: 1, 180: 0, 350: 1, 520: 0})
    run_scripted("HarfangCircularEnv", 3, 700, True, {90: 0, 180: 1, 350: 0, 520: 1})
    run_scripted("HarfangSerpentineInfiniteEnv", 4, 700, True, {})
    np.savez_compressed(os.path.join(HERE, "env_opponent_stream.npz"),
                        serpentine=opponent_stream("HarfangSerpentineEnv", 2000),
                        circular=opponent_stream("HarfangCircularEnv", 2000),
                        straight_line=opponent_stream("HarfangEnv", 300))
    run_closedloop("HarfangEnv", 0, 1500, "straight_line")
    run_closedloop("HarfangSerpentineEnv", 1, 1500, "serpentine")
    run_closedloop("HarfangCircularEnv", 2, 1900, "circular")
    gen_getreward()
    gen_random_reset()
