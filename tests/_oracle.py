"""ctypes bindings of the CPU oracle (oracle/libhx_oracle.so) — test infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# HX_ORACLE_LIBRARY selects another build of the same restatement (oracle/Makefile `asan`: tools/asan_oracle.sh)
_SO = os.environ.get("HX_ORACLE_LIBRARY") or os.path.join(REPO, "oracle", "libhx_oracle.so")

F_LOCKED_PREV, F_LOCKED, F_SLOT_PREV, F_SLOT, F_FIRED, F_FIRE_SUCCESS, F_EPISODE_SUCCESS, F_DONE = (1 << i for i in range(8))
F_SCEN_SHIFT = 8
F_SERP_POS, F_SERP_LONG, F_M_ACTIVE, F_M_GUIDED, F_SIM_SLOT = (1 << i for i in range(10, 15))

_vp = ctypes.c_void_p
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")])
        L = ctypes.CDLL(_SO)
        L.ox_env_reset.argtypes = [_vp, ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32]
        L.ox_env_observe.argtypes = [_vp, _vp]
        L.ox_env_step.argtypes = [_vp] * 6
        L.ox_env_rearm.argtypes = [_vp]
        L.ox_sim_tick.argtypes = [_vp, _vp, _vp, ctypes.c_int]
        L.ox_sim_readback.argtypes = [_vp, _vp]
        L.ox_wrap_observe.argtypes = [_vp, _vp, _vp]
        L.ox_wrap_reward.argtypes = [_vp, _vp, _vp]
        L.ox_wrap_reward.restype = ctypes.c_float
        L.ox_wrap_terminate.argtypes = [_vp, _vp]
        L.ox_script_opponent.argtypes = [_vp, _vp, _vp]
        L.ox_get_reward.argtypes = [_vp] * 4
        L.ox_get_reward.restype = ctypes.c_float
        L.ox_get_termination.argtypes = [_vp]
        L.ox_philox4x32_10.argtypes = [_vp] * 3
        L.ox_env_step_batch.argtypes = [_vp, ctypes.c_int64, _vp, _vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_uint64, ctypes.c_uint32, _vp, _vp, _vp, ctypes.c_int64, _vp, _vp]
        for f in (L.ox_asin, L.ox_acos):
            f.argtypes, f.restype = [ctypes.c_float], ctypes.c_float
        L.ox_atan2.argtypes, L.ox_atan2.restype = [ctypes.c_float, ctypes.c_float], ctypes.c_float
        assert L.ox_sizeof_env() == 148
        _lib = L
    return _lib


def p(a):
    return a.ctypes.data if a is not None else None


def make_readback(row16):
    """fixture row [ally pos3 eul3, opp pos3 eul3, angle, health, locked(0/1), slot(0/1)] -> OxReadback bytes"""
    rb = np.zeros(16, np.float32)
    rb[:14] = row16[:14]
    rb[14:16].view(np.int32)[:] = (int(row16[14]), int(row16[15]))
    return rb


def reset_batch(n, scenario, randomize, seed, env_id0=0, episode=0):
    """-> envs [n, 37] float32 (AoS, oracle layout) and obs [n, 13]"""
    L = lib()
    envs = np.zeros((n, 37), np.float32)
    obs = np.zeros((n, 13), np.float32)
    scen = np.broadcast_to(np.asarray(scenario), (n,))
    for i in range(n):
        L.ox_env_reset(envs[i].ctypes.data, int(scen[i]), int(randomize), int(seed), env_id0 + i, episode)
        L.ox_env_observe(envs[i].ctypes.data, obs[i].ctypes.data)
    return envs, obs


def step_batch(envs, actions, obs_io, max_step=0, auto_reset=0, randomize=0, seed=0, env_id0=0, episode_ctr=None,
               ring=None, ring_succ=None, total=None, stats=None):
    L = lib()
    n = envs.shape[0]
    reward = np.zeros(n, np.float32)
    done = np.zeros(n, np.uint8)
    succ = np.zeros(n, np.int8)
    if episode_ctr is None:
        episode_ctr = np.zeros(n, np.uint32)
    cap = ring.shape[0] if ring is not None else 0
    L.ox_env_step_batch(p(envs), n, p(np.ascontiguousarray(actions, np.float32)), p(obs_io), p(reward), p(done), p(succ),
                        int(max_step), int(auto_reset), int(randomize), int(seed), int(env_id0), p(episode_ctr),
                        p(ring), p(ring_succ), cap, p(total), p(stats))
    return reward, done, succ
