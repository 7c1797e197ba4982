"""BatchedHarfangEnv — N pursuit-lock-launch envs resident on one MI355X, stepped by one HIP launch.

Vectorised counterpart of the reference's single-socket HarfangEnv (hirl/environments/HarfangEnv_GYM.py): same
observation / action / reward / done / success contract per env, with the driver's episode rules
(train_all.py:341-361) applied inside the kernel when auto_reset is on.  Every method only enqueues work on the
current torch stream; returned tensors are views of buffers owned by this object.
"""
import ctypes

import torch

from .. import _lib

SCENARIOS = {"straight_line": 0, "serpentine": 1, "circular": 2}


class BatchedHarfangEnv:
    def __init__(self, num_envs, scenario="straight_line", device="cuda", seed=0, max_step=0, auto_reset=True,
                 random_reset=True, env_id0=0, replay=None, collect_stats=True, layout=0, pitch=0):
        self.n = int(num_envs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HxError("BatchedHarfangEnv runs on the GPU only (there is no CPU path in the product)")
        _lib.load()
        self.seed, self.env_id0 = int(seed), int(env_id0)
        self.max_step, self.auto_reset, self.random_reset = int(max_step), bool(auto_reset), bool(random_reset)
        if isinstance(scenario, str):
            self.scenario_all, self.scenario = SCENARIOS[scenario], None
        elif isinstance(scenario, int):
            self.scenario_all, self.scenario = scenario, None
        else:  # per-env ids; the mixed config sorts envs by scenario so that wavefronts do not diverge
            self.scenario_all = 0
            self.scenario = torch.as_tensor(scenario, dtype=torch.int32, device=self.device).contiguous()
        d = self.device
        # struct-of-arrays state: word w of env i at state[w, i].  pitch (>= n, floats) is the distance between two words' arrays — the
        # C ABI's `stride`.  0 = the default: n, and n + 1,056 floats from 262,144 envs on — with the arrays a power of two apart the 37
        # streams of a launch lean on the same HBM channels: 1M envs 105.8 -> 102.6 us per launch (tools/ubench/env_pitch.py; nothing at
        # 65,536 or 4M envs).
        self.pitch = max(int(pitch), self.n) if pitch else (self.n + 1056 if self.n >= 262144 else self.n)
        self._state_store = torch.zeros((_lib.ENV_WORDS, self.pitch), dtype=torch.float32, device=d)
        self.state = self._state_store[:, :self.n]
        self.obs = torch.zeros((self.n, _lib.OBS_DIM), dtype=torch.float32, device=d)
        self.reward = torch.zeros(self.n, dtype=torch.float32, device=d)
        self.done = torch.zeros(self.n, dtype=torch.uint8, device=d)
        self.success = torch.zeros(self.n, dtype=torch.int8, device=d)
        self.episode_ctr = torch.zeros(self.n, dtype=torch.int32, device=d)
        self.stats = torch.zeros((_lib.STAT_WAYS, _lib.STAT_PITCH), dtype=torch.int64, device=d) if collect_stats else None  # sum over dim 0: stats_dict()
        self.replay = replay
        self.steps_issued = 0  # step launches so far (every path that may insert into `replay`): HirlEngine.step_learn checks its ring snapshot against it
        self.layout = int(layout)  # 0: the library picks the launch shape; _lib.layout(pair, envs_per_block) forces one (tests, tuning)
        self._opts = _lib.HxStepOpts()
        self._refresh_opts()

    def _refresh_opts(self):
        o = self._opts
        o.max_step, o.auto_reset, o.randomize = self.max_step, int(self.auto_reset), int(self.random_reset)
        o.env_id0, o.seed = self.env_id0, self.seed
        o.episode_ctr = _lib.ptr(self.episode_ctr)
        r = self.replay
        o.ring, o.ring_success = (_lib.ptr(r.ring), _lib.ptr(r.success)) if r is not None else (None, None)
        o.cap, o.total = (r.capacity, _lib.ptr(r.total)) if r is not None else (0, None)
        o.stats = _lib.ptr(self.stats)
        o.layout = self.layout

    def reset(self, mask=None):
        """reset() / random_reset() of every env (or of the envs with mask != 0) -> obs [N, 13]."""
        if mask is not None:
            mask = mask.to(self.device, torch.uint8).contiguous()
        _lib.call("hx_env_reset", _lib.ptr(self.state), self.n, self.pitch, _lib.ptr(mask), _lib.ptr(self.scenario),
                  self.scenario_all, int(self.random_reset), self.seed, self.env_id0, _lib.ptr(self.episode_ctr),
                  _lib.ptr(self.obs), _lib.stream_ptr())
        return self.obs

    def step(self, actions):
        """actions [N, 4] fp32 on the device -> (obs [N,13], reward [N], done [N] u8, success [N] i8)."""
        if actions.dtype != torch.float32 or not actions.is_contiguous() or actions.device != self.device:
            actions = actions.to(self.device, torch.float32).contiguous()
        self.steps_issued += 1
        _lib.call("hx_env_step", _lib.ptr(self.state), self.n, self.pitch, _lib.ptr(actions), _lib.ptr(self.obs),
                  _lib.ptr(self.reward), _lib.ptr(self.done), _lib.ptr(self.success), ctypes.byref(self._opts),
                  _lib.stream_ptr())
        return self.obs, self.reward, self.done, self.success

    def step_from(self, actions_ptr):
        """step() with the actions at a raw address the DEVICE can read — device memory, or pinned host memory it maps (the N = 1 facade
        passes its pinned action buffer: no upload call)."""
        self.steps_issued += 1
        _lib.call("hx_env_step", _lib.ptr(self.state), self.n, self.pitch, actions_ptr, _lib.ptr(self.obs),
                  _lib.ptr(self.reward), _lib.ptr(self.done), _lib.ptr(self.success), ctypes.byref(self._opts),
                  _lib.stream_ptr())
        return self.obs, self.reward, self.done, self.success

    def time_next_steps(self, start=None, stop=None):
        """Measurement: hx_event_create() handles that the following step launches stamp with the kernel's own begin / end
        (None, None switches it off).  Read with hx_event_elapsed_us after a synchronise."""
        self._opts.ev_start, self._opts.ev_stop = getattr(start, "value", start), getattr(stop, "value", stop)

    def rearm(self, mask=None):
        _lib.call("hx_env_rearm", _lib.ptr(self.state), self.n, self.pitch, _lib.ptr(mask), _lib.stream_ptr())

    def stats_dict(self):
        if self.stats is None:
            return {}
        return dict(zip(_lib.STAT_NAMES, (int(v) for v in self.stats.sum(0).tolist())))

    # raw state access for parity tests (the C ABI leaves the state buffer with the caller) -------------------
    def get_state(self):
        return self.state.clone()

    def set_state(self, state, obs=None):
        self.state.copy_(state.to(self.device, torch.float32).reshape(_lib.ENV_WORDS, self.n))
        if obs is not None:
            self.obs.copy_(obs.to(self.device, torch.float32))
