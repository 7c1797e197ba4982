"""Drop-in counterparts of the reference env classes (hirl/environments/HarfangEnv_GYM.py) for N = 1: same
constructor (no arguments), same reset / random_reset / step / step_test tuples, same public attributes — backed by
the batched HIP env-step kernel (one env resident on the GPU) instead of a Harfang sandbox over TCP.

For throughput use BatchedHarfangEnv (environments/batched.py); this façade exists so that the reference's drivers
(train_all.py, validate_all.py) and data tools keep working call for call.
"""
import ctypes
import inspect
import os
import random

import numpy as np
import torch

from .. import _lib
from .batched import BatchedHarfangEnv
from .constants import *  # noqa: F401,F403  (the reference star-exports NormStates the same way, HarfangEnv_GYM.py:3)


class _Box:
    """The one thing the reference uses gym.spaces.Box for: action_space.sample() (train_all.py:272)."""

    def __init__(self, low, high, dtype=np.float64):
        self.low, self.high, self.dtype = np.asarray(low, dtype), np.asarray(high, dtype), dtype
        self.shape = self.low.shape

    def sample(self):
        return np.random.uniform(self.low, self.high).astype(self.dtype)


class HarfangEnv:
    _scenario = 0

    def __init__(self):  # HarfangEnv_GYM.py:11-32
        self.done = False
        self.loc_diff = 0
        self.action_space = _Box(low=[-1.0, -1.0, -1.0, -1.0], high=[1.0, 1.0, 1.0, 1.0], dtype=np.float64)
        self.Plane_ID_oppo = "ennemy_2"
        self.Plane_ID_ally = "ally_1"
        self.Aircraft_Loc = None
        self.Ally_target_locked = False
        self.n_Ally_target_locked = False
        self.reward = 0
        self.Plane_Irtifa = 0
        self.now_missile_state = False
        self.missile1_state = True
        self.n_missile1_state = True
        self.oppo_health = 0.2
        self.target_angle = None
        self.success = 0
        self.episode_success = False
        self.fire_success = False
        # the reference draws reset offsets from Python's global `random`; keep that dependency so set_seed() still
        # makes runs repeatable: the Philox key is taken from it once
        self._env = BatchedHarfangEnv(1, scenario=self._scenario, seed=random.getrandbits(63), auto_reset=False,
                                      random_reset=False, collect_stats=False)
        # host <-> device traffic of a step: the kernels read the action from, and hx_env_pack_row writes the read-back row (state words,
        # observation, reward, done, success: 64 floats) into, PINNED HOST memory the device maps — no copy calls, one synchronisation
        self._h_action = torch.zeros((1, 4), dtype=torch.float32).pin_memory()
        self._h_row = torch.zeros(64, dtype=torch.float32).pin_memory()
        self._np_action, self._np_row = self._h_action.numpy(), self._h_row.numpy()
        # (the stream current AT THE CALL: hx_env_step / hx_env_pack_row are enqueued on _lib.stream_ptr(), which follows torch.cuda.stream(...))
        self._stream_sync = lambda: torch.cuda.current_stream(self._env.device).synchronize()
        _lib.register("hx_env_pack_row", [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p])
        self.state = None

    # ---- helpers ------------------------------------------------------------------------------------------------
    def _readback(self):
        """-> (state words [37], obs [13] float64, reward, success) of the one env, through the packed row"""
        e = self._env
        _lib.call("hx_env_pack_row", e.state.data_ptr(), 1, e.pitch, 0, e.obs.data_ptr(), e.reward.data_ptr(), e.done.data_ptr(), e.success.data_ptr(),
                  self._h_row.data_ptr(), _lib.stream_ptr())
        self._stream_sync()
        row = self._np_row
        return row[0:37].copy(), row[37:50].astype(np.float64), float(row[50]), int(row[52])

    def _pull(self, st=None):
        if st is None:
            st = self._env.state[:, 0].cpu().numpy()
        flags = int(st.view(np.uint32)[35])
        self.Aircraft_Loc = st[0:3].tolist()
        self.Oppo_Loc = st[13:16].tolist()
        self.Plane_Irtifa = self.Aircraft_Loc[1]
        self.Ally_target_locked = bool(flags & _lib.F_LOCKED_PREV)
        self.n_Ally_target_locked = bool(flags & _lib.F_LOCKED)
        self.missile1_state = bool(flags & _lib.F_SLOT_PREV)
        self.n_missile1_state = bool(flags & _lib.F_SLOT)
        self.now_missile_state = bool(flags & _lib.F_FIRED)
        self.episode_success = bool(flags & _lib.F_EPISODE_SUCCESS)
        self.fire_success = bool(flags & _lib.F_FIRE_SUCCESS)
        self.done = bool(flags & _lib.F_DONE)
        self.oppo_health = {"health_level": float(st[32])}
        d = st[0:3].astype(np.float64) - st[13:16].astype(np.float64)
        self.loc_diff = float(np.sqrt((d * d).sum()))  # HarfangEnv_GYM.py:190-191

    def _do_reset(self, randomize):
        self._env.random_reset = bool(randomize)
        self._env.episode_ctr += 1  # a fresh Philox counter per episode
        self._env.reset()
        st, o, _, _ = self._readback()
        self._pull(st)
        self.target_angle = float(o[6])
        self.success = 0
        self.state = o
        return o

    # ---- reference API ------------------------------------------------------------------------------------------
    def reset(self):  # HarfangEnv_GYM.py:34-49
        return self._do_reset(False)

    def random_reset(self):  # :51-66
        return self._do_reset(True)

    def _step(self, action):
        self._np_action[0] = (float(action[0]), float(action[1]), float(action[2]), float(action[3]))
        self._env.step_from(self._h_action.data_ptr())  # (pinned host memory: the env kernel reads its 16 bytes over the host link)
        st, o, self.reward, self.success = self._readback()
        self._pull(st)
        self.target_angle = float(o[6])
        self.state = o
        return o

    def step(self, action):  # :83-90
        n_state = self._step(action)
        return n_state, self.reward, self.done, {}, self.success

    def step_test(self, action):  # :92-99
        n_state = self._step(action)
        return (n_state, self.reward, self.done, {}, self.now_missile_state, self.missile1_state, self.n_missile1_state,
                self.Ally_target_locked, self.success)

    def get_pos(self):  # :270-274  (the positions the last read-back cached: no device access)
        return np.asarray(self.Aircraft_Loc, np.float64)

    def get_oppo_pos(self):  # :276-280
        return np.asarray(self.Oppo_Loc, np.float64)

    def save_parameters_to_txt(self, log_dir):  # :282-295 (dumps the source of the reward / reset / termination rules)
        with open(os.path.join(log_dir, "log2.txt"), "w") as f:
            f.write(inspect.getsource(type(self)))

    # ---- expert-data labelling (pure functions of the observation) ---------------------------------------------
    def get_loc_diff(self, state):  # :299-301
        return (((state[0]) * 10000) ** 2 + ((state[1]) * 10000) ** 2 + ((state[2]) * 10000) ** 2) ** (1 / 2)

    def get_reward(self, state, action, n_state):  # :303-330
        reward = 0
        step_success = 0
        reward -= 0.0001 * self.get_loc_diff(n_state)
        reward -= (n_state[6]) * 10
        if action[-1] > 0:
            reward -= 8
            if state[8] > 0 and state[7] < 0:
                step_success = -1
            elif state[8] > 0 and state[7] > 0:
                step_success = 1
        if n_state[-1] < 0.1:
            reward += 600
        return reward, step_success

    def get_termination(self, state):  # :332-336
        return bool(state[-1] <= 0.1)


class HarfangSerpentineEnv(HarfangEnv):  # HarfangEnv_GYM.py:338-406 — opponent yaw script runs inside the kernel
    _scenario = 1


class HarfangCircularEnv(HarfangEnv):  # :408-474
    _scenario = 2


class HarfangSerpentineInfiniteEnv(HarfangSerpentineEnv):  # :476-535
    def __init__(self):
        super().__init__()
        self.infinite_total_success = 0
        self.infinite_total_fire = 0
        self.infinite_total_step = 0

    def step_test(self, action):
        self.infinite_total_step += 1
        if self.infinite_total_step % 60 == 0:  # :484-486
            self._env.rearm()
        out = super().step_test(action)
        if self.success != 0:  # :519,525-526
            self.infinite_total_fire += 1
        if self.success == 1:
            self.infinite_total_success += 1
        return out
