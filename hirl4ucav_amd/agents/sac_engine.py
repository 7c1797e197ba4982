"""SacEngine — device-resident state of the SAC agent (reference hirl/agents/SAC/agent.py, non-imitative branch) and the
host-side sequencing of the hx_sac_* stages.  The networks are the plain Linear-ReLU stacks of the reference's (un-vendored)
rltorch builder; they reuse the flat MLP-block layout of include/hirl4ucav.h with the LayerNorm slots pinned to (1, 0)."""
import ctypes
import os

import numpy as np
import torch

from .. import _lib
from . import engine as E

_vp, _i32, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float
_P = ctypes.POINTER


class HxSacNets(ctypes.Structure):
    _fields_ = [(k, _vp) for k in ("policy", "critic", "target_critic", "grad_policy", "grad_critic", "m_policy", "v_policy", "m_critic",
                                   "v_critic", "losses", "alpha_state", "ws", "policy_w2_f32i", "policy_w2_x9")]


class HxSacBatch(ctypes.Structure):
    _fields_ = [("rows", _vp), ("batch", _i32), ("eps_next", _vp), ("eps_cur", _vp), ("seed", ctypes.c_uint64), ("call", ctypes.c_uint32)]


_lib.register("hx_sac_act", [_vp, _vp, ctypes.c_int64, _vp, _i32, _vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _vp, _vp])
_lib.register("hx_sac_act_step", [_vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _i32, _vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                   _vp, _vp, _vp, _P(_lib.HxStepOpts), _vp])
_lib.register("hx_sac_act_f32i", [_vp, _vp, _vp, ctypes.c_int64, _vp, _i32, _vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _vp])
_lib.register("hx_sac_act_step_f32i", [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _i32, _vp, ctypes.c_uint64, ctypes.c_uint32,
                                        ctypes.c_uint32, _vp, _vp, _vp, _P(_lib.HxStepOpts), _vp])
_lib.register("hx_sac_act_x9", [_vp, _vp, _vp, _vp, ctypes.c_int64, _vp, _i32, _vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _vp])
_lib.register("hx_sac_act_step_x9", [_vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _i32, _vp, ctypes.c_uint64, ctypes.c_uint32,
                                      ctypes.c_uint32, _vp, _vp, _vp, _P(_lib.HxStepOpts), _vp])
_lib.register("hx_sac_critic_grads", [_P(HxSacNets), _P(HxSacBatch), _P(E.HxHyper), _i32, _vp])
_lib.register("hx_sac_critic_grads_sampled", [_P(HxSacNets), _P(HxSacBatch), _P(E.HxHyper), _P(E.HxSample), _i32, _vp])
_lib.register("hx_sac_critic_step", [_P(HxSacNets), _P(HxSacBatch), _P(E.HxHyper), _P(E.HxSample), _i32, _i32, _vp])
_lib.register("hx_sac_front", [_vp, _vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _i32, _vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _vp, _vp, _vp,
                                _P(_lib.HxStepOpts), _P(HxSacNets), _P(HxSacBatch), _vp])
_lib.register("hx_sac_learn_back", [_P(HxSacNets), _P(HxSacBatch), _P(E.HxHyper), _i32, _i32, ctypes.c_float, _P(E.HxSample), _vp, _vp])
_lib.register("hx_sac_learn", [_P(HxSacNets), _P(HxSacBatch), _P(E.HxHyper), _P(E.HxSample), _i32, _i32, ctypes.c_float, _vp])
_lib.register("hx_sac_policy_grads", [_P(HxSacNets), _P(HxSacBatch), _P(E.HxHyper), _vp])
_lib.register("hx_sac_adam", [_P(HxSacNets), _P(E.HxHyper), _i32, _i32, _f32, _f32, _vp])

H1, H2 = E.H1, E.H2
SEQ_KEYS = ("0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias")  # nn.Sequential(Linear, ReLU, Linear, ReLU, Linear)


def _block(in_dim, out_dim):
    """offsets of W1 b1 g1 be1 W2 b2 g2 be2 W3 b3 inside one block (hx_nn.h Mlp)"""
    o, off = {}, 0
    for k, n in (("W1", H1 * in_dim), ("b1", H1), ("g1", H1), ("be1", H1), ("W2", H2 * H1), ("b2", H2), ("g2", H2), ("be2", H2),
                 ("W3", out_dim * H2), ("b3", out_dim)):
        o[k] = (off, n)
        off += n
    return o, (off + 3) & ~3


POLICY_BLOCK, POLICY_SIZE = _block(13, 8)
Q_BLOCK, Q_SIZE = _block(17, 1)
assert Q_SIZE == E.Q_PADDED


def pack_mlp(sd, block, size, in_dim, out_dim, device):
    """Sequential state_dict -> flat block (LayerNorm slots = (1, 0))"""
    flat = torch.zeros(size, dtype=torch.float32, device=device)
    shapes = {"0.weight": (H1, in_dim), "0.bias": (H1,), "2.weight": (H2, H1), "2.bias": (H2,), "4.weight": (out_dim, H2), "4.bias": (out_dim,)}
    for key, slot in zip(SEQ_KEYS, ("W1", "b1", "W2", "b2", "W3", "b3")):
        v = torch.as_tensor(np.asarray(sd[key].detach().cpu() if torch.is_tensor(sd[key]) else sd[key]), dtype=torch.float32)
        assert tuple(v.shape) == shapes[key], (key, v.shape)
        off, n = block[slot]
        flat[off:off + n] = v.reshape(-1).to(device)
    for slot in ("g1", "g2"):
        off, n = block[slot]
        flat[off:off + n] = 1.0
    return flat


def unpack_mlp(flat, block, in_dim, out_dim):
    shapes = {"W1": (H1, in_dim), "b1": (H1,), "W2": (H2, H1), "b2": (H2,), "W3": (out_dim, H2), "b3": (out_dim,)}
    return {key: flat[block[slot][0]:block[slot][0] + block[slot][1]].reshape(shapes[slot])
            for key, slot in zip(SEQ_KEYS, ("W1", "b1", "W2", "b2", "W3", "b3"))}


class SacEngine:
    def __init__(self, batch=128, lr=1e-3, gamma=0.99, tau=0.005, target_entropy=-4.0, target_update_interval=3, device="cuda", group=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HxError("SacEngine runs on the GPU only (no CPU path in the product)")
        L = _lib.load()
        L.hx_sac_workspace_floats.restype = ctypes.c_int64
        assert L.hx_sac_policy_param_count() == POLICY_SIZE
        self.batch = int(batch)
        ws = int(L.hx_sac_workspace_floats(self.batch))
        sizes = [POLICY_SIZE, 2 * Q_SIZE, 2 * Q_SIZE, POLICY_SIZE + 2 * Q_SIZE, POLICY_SIZE, POLICY_SIZE, 2 * Q_SIZE, 2 * Q_SIZE, 64, 64, ws,
                 self.batch * 32, self.batch * 4, self.batch * 4, self.batch, 64]
        offs, tot = [], 0
        for n in sizes:
            offs.append(tot)
            tot += (n + 63) & ~63
        self.arena = torch.zeros(tot, dtype=torch.float32, device=self.device)
        c = [self.arena[o:o + n] for o, n in zip(offs, sizes)]
        (self.policy, self.critic, self.target_critic, self.grad, self.m_policy, self.v_policy, self.m_critic, self.v_critic, losses,
         alpha, self.ws, self.rows, self.eps_next, self.eps_cur, ix, noise) = c
        self.grad_critic, self.grad_policy = self.grad[:2 * Q_SIZE], self.grad[2 * Q_SIZE:]
        self.losses, self.alpha_state, self._idx, self._noise = losses[:8], alpha[:4], ix.view(torch.int32), noise[:4]
        self.alpha_state[3] = 1.0  # log_alpha = 0 -> alpha = 1  (agent.py:106-107)
        self.nets = HxSacNets(*(t.data_ptr() for t in (self.policy, self.critic, self.target_critic, self.grad_policy, self.grad_critic,
                                                        self.m_policy, self.v_policy, self.m_critic, self.v_critic, self.losses,
                                                        self.alpha_state, self.ws)), None, None)
        for i, cls in ((5, HxSacNets), (6, HxSacBatch), (0, _lib.HxStepOpts), (2, E.HxHyper), (4, E.HxSample)):
            _lib.check_struct(i, cls)  # this binding's structs against the loaded library's (hx_abi_sizes)
        # fp32 image of the policy's W2 in the acting kernel's operand order (hx_pack_w2_f32i): kept current by hx_sac_adam(which = 1)
        self.w2_f32i = torch.zeros(512 * 256, dtype=torch.float32, device=self.device)
        self.nets.policy_w2_f32i = self.w2_f32i.data_ptr()
        # From this many rows on the policy's fp32 256 -> 512 product runs through the exact three-way bf16 split of both operands on the bf16 matrix cores
        # (hx_sac_act*_x9: every partial product exact, fp32 accumulation; 34 against 45 us at 16,384 rows, tools/ubench/actp_time.py);
        # None: fp32 MFMA at every size.  The hi | mid | lo images are built at the first such call, the policy's optimizer step keeps them current.
        self.x9_rows, self.w2_x9 = 16384, None
        self.hyper = E.HxHyper(gamma, tau, lr, lr, 0.0, 0.5, 0.0, 0)
        self.target_entropy, self.interval = float(target_entropy), int(target_update_interval)
        self.learning_steps = 0
        self.group = group
        self.world = torch.distributed.get_world_size(group) if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
        self.act_calls, self.sample_calls = 0, 0

    def load_params(self, policy, q1, q2, hard_update_target=True):
        self.policy.copy_(pack_mlp(policy, POLICY_BLOCK, POLICY_SIZE, 13, 8, self.device))
        self.critic[:Q_SIZE].copy_(pack_mlp(q1, Q_BLOCK, Q_SIZE, 17, 1, self.device))
        self.critic[Q_SIZE:].copy_(pack_mlp(q2, Q_BLOCK, Q_SIZE, 17, 1, self.device))
        if hard_update_target:  # hard_update(critic_target, critic), agent.py:92
            self.target_critic.copy_(self.critic)
        self.refresh_images()

    def refresh_images(self):
        """Rebuild the acting kernel's image of the policy's W2 (after load_params / a checkpoint restore or any direct write to
        `self.policy`; the policy's Adam step maintains it otherwise)."""
        _lib.call("hx_pack_w2_f32i", self.policy.data_ptr(), 13, self.w2_f32i.data_ptr(), _lib.stream_ptr())
        if self.w2_x9 is not None:
            _lib.call("hx_pack_w2_x9", self.policy.data_ptr(), 13, self.w2_x9.data_ptr(), _lib.stream_ptr())

    def _x9_for(self, n):
        if self.x9_rows is None or n < self.x9_rows:
            return False
        if self.w2_x9 is None:  # first large call: build the images; nets.policy_w2_x9 makes every later policy step refresh them
            self.w2_x9 = torch.zeros(3 * H2 * H1, dtype=torch.bfloat16, device=self.device)
            self.nets.policy_w2_x9 = self.w2_x9.data_ptr()
            _lib.call("hx_pack_w2_x9", self.policy.data_ptr(), 13, self.w2_x9.data_ptr(), _lib.stream_ptr())
        return True

    refresh_bf16 = refresh_images  # (the name utils/checkpoint.py calls after a restore)

    def replica_checksum(self):
        """int64 sum of the bit patterns of the networks, Adam moments and log-alpha state: equal on all ranks of a sharded run."""
        t = torch.cat([self.policy, self.critic, self.target_critic, self.m_policy, self.v_policy, self.m_critic, self.v_critic, self.alpha_state])
        return int(t.view(torch.int32).to(torch.int64).sum().item())

    def state_dicts(self):
        return {"policy": unpack_mlp(self.policy, POLICY_BLOCK, 13, 8), "q1": unpack_mlp(self.critic[:Q_SIZE], Q_BLOCK, 17, 1),
                "q2": unpack_mlp(self.critic[Q_SIZE:], Q_BLOCK, 17, 1), "q1_target": unpack_mlp(self.target_critic[:Q_SIZE], Q_BLOCK, 17, 1),
                "q2_target": unpack_mlp(self.target_critic[Q_SIZE:], Q_BLOCK, 17, 1)}

    def model_files(self):
        """What SacAgent.save_models writes (SAC/agent.py:440-444): the state_dicts of GaussianPolicy, TwinnedQNetwork and its
        target with the reference's key names (SAC/model.py:21,35-38,58: `policy.N.*`, `Q1.Q.N.*`, `Q2.Q.N.*`)."""
        sd = self.state_dicts()
        cpu = lambda d, pre: {pre + k: v.cpu().clone() for k, v in d.items()}
        return {"policy": cpu(sd["policy"], "policy."),
                "critic": {**cpu(sd["q1"], "Q1.Q."), **cpu(sd["q2"], "Q2.Q.")},
                "critic_target": {**cpu(sd["q1_target"], "Q1.Q."), **cpu(sd["q2_target"], "Q2.Q.")}}

    def save_models(self, model_dir, tag):
        for name, sd in self.model_files().items():
            torch.save(sd, os.path.join(model_dir, f"{name}_{tag}.pth"))

    def load_models(self, model_dir, tag):
        strip = lambda d, pre: {k[len(pre):]: v for k, v in d.items() if k.startswith(pre)}
        f = {name: torch.load(os.path.join(model_dir, f"{name}_{tag}.pth"), map_location="cpu") for name in ("policy", "critic", "critic_target")}
        self.load_params(strip(f["policy"], "policy."), strip(f["critic"], "Q1.Q."), strip(f["critic"], "Q2.Q."), hard_update_target=False)
        self.target_critic[:Q_SIZE].copy_(pack_mlp(strip(f["critic_target"], "Q1.Q."), Q_BLOCK, Q_SIZE, 17, 1, self.device))
        self.target_critic[Q_SIZE:].copy_(pack_mlp(strip(f["critic_target"], "Q2.Q."), Q_BLOCK, Q_SIZE, 17, 1, self.device))

    def act(self, obs, eps=None, explore=True, seed=0, row0=0, out=None):
        """explore (agent.py:183-188): sampled tanh-Gaussian action (eps [N, 4] given, else Philox); exploit (:191-196): tanh(mean)."""
        n = obs.shape[0]
        if out is None:
            out = torch.empty((n, 4), dtype=torch.float32, device=self.device)
        mode = 0 if not explore else (1 if eps is not None else 2)
        self.act_calls += 1
        if self._x9_for(n):
            _lib.call("hx_sac_act_x9", self.policy.data_ptr(), self.w2_x9.data_ptr(), self.w2_f32i.data_ptr(), obs.data_ptr(), n, out.data_ptr(), mode,
                      _lib.ptr(eps), int(seed), int(row0), self.act_calls, _lib.stream_ptr())
            return out
        _lib.call("hx_sac_act_f32i", self.policy.data_ptr(), self.w2_f32i.data_ptr(), obs.data_ptr(), n, out.data_ptr(), mode, _lib.ptr(eps),
                  int(seed), int(row0), self.act_calls, _lib.stream_ptr())
        return out

    def act_step(self, env, eps=None, explore=True, seed=0, out=None):
        """explore / exploit for every env of `env` AND env.step with those actions in one launch (train_sac.py:238-241); same
        results as act(env.obs, ...) followed by env.step(actions).  -> (actions, obs, reward, done, success)."""
        n = env.n
        if out is None:
            out = torch.empty((n, 4), dtype=torch.float32, device=self.device)
        mode = 0 if not explore else (1 if eps is not None else 2)
        self.act_calls += 1
        env.steps_issued += 1
        if self._x9_for(n):
            _lib.call("hx_sac_act_step_x9", self.policy.data_ptr(), self.w2_x9.data_ptr(), self.w2_f32i.data_ptr(), env.state.data_ptr(), n, env.pitch,
                      env.obs.data_ptr(), out.data_ptr(), mode, _lib.ptr(eps), int(seed), int(env.env_id0), self.act_calls, env.reward.data_ptr(),
                      env.done.data_ptr(), env.success.data_ptr(), ctypes.byref(env._opts), _lib.stream_ptr())
            return out, env.obs, env.reward, env.done, env.success
        _lib.call("hx_sac_act_step_f32i", self.policy.data_ptr(), self.w2_f32i.data_ptr(), env.state.data_ptr(), n, env.pitch, env.obs.data_ptr(),
                  out.data_ptr(), mode, _lib.ptr(eps), int(seed), int(env.env_id0), self.act_calls, env.reward.data_ptr(), env.done.data_ptr(),
                  env.success.data_ptr(), ctypes.byref(env._opts), _lib.stream_ptr())
        return out, env.obs, env.reward, env.done, env.success

    def assemble(self, ring, idx):
        _lib.call("hx_sample_batch", None, 0, ring.data_ptr(), None, 0, None, 0, self.batch, self.batch, 0, 0, 0, 0.0, idx.data_ptr(), None,
                  None, self.rows.data_ptr(), None, _lib.stream_ptr())

    def sample(self, replay, expert=None, n_main=None, seed=0, defer=False):
        """memory.sample(batch_size) on the device (E-SAC: batch - expert_num rows from the memory followed by expert_num rows of
        the expert memory, SAC/agent.py:286-296, train_sac.py:401-403) + the two standard-normal draw sets of the learn() call.
        defer=True: nothing is launched now — the next learn() draws and gathers inside its first launch (hx_sac_critic_grads_sampled)."""
        self.sample_calls += 1
        n_main = self.batch if (n_main is None or expert is None) else int(n_main)
        self._seed = int(seed)
        if defer:
            self._pending = (E.HxSample(replay.total.data_ptr(), replay.capacity, replay.ring.data_ptr(),
                                        expert.ring.data_ptr() if expert is not None else None, E.len_of(expert), None, 0, n_main, int(seed),
                                        self.sample_calls, 0.0, self._idx.data_ptr(), None), replay, expert)
            return
        self._pending = None
        _lib.call("hx_sample_batch", replay.total.data_ptr(), replay.capacity, replay.ring.data_ptr(),
                  expert.ring.data_ptr() if expert is not None else None, E.len_of(expert), None, 0, self.batch, n_main,
                  1, int(seed), self.sample_calls, 0.0, self._idx.data_ptr(), None, self._noise.data_ptr(), self.rows.data_ptr(), None,
                  _lib.stream_ptr())
        self._seed = int(seed)  # learn() without injected draws: Normal.rsample's eps comes from Philox(seed; row, learn call) in-kernel

    def _allreduce(self, t):
        if self.world > 1:
            torch.distributed.all_reduce(t, group=self.group)

    def learn(self, eps_next=None, eps_cur=None):
        """SacAgent.learn(False) (agent.py:276-327) on the minibatch last assembled.  Enqueues only."""
        st = _lib.stream_ptr()
        if eps_next is not None:  # injected draws (parity tests, the façade's torch.randn)
            self.eps_next.copy_(eps_next.reshape(-1))
            self.eps_cur.copy_(eps_cur.reshape(-1))
            batch = HxSacBatch(self.rows.data_ptr(), self.batch, self.eps_next.data_ptr(), self.eps_cur.data_ptr(), 0, 0)
        else:
            batch = HxSacBatch(self.rows.data_ptr(), self.batch, None, None, getattr(self, "_seed", 0), self.learning_steps + 1)
        nets, hyper, gs = ctypes.byref(self.nets), ctypes.byref(self.hyper), 1.0 / self.world
        self.learning_steps += 1
        pending, self._pending = getattr(self, "_pending", None), None
        polyak_first = int(self.learning_steps % self.interval == 0)
        if self.world == 1 and not getattr(self, "separate_critic_adam", False) and not getattr(self, "staged_policy", False):
            # one GPU: the whole learn() in one call, 9 launches (hx_sac_learn: bit-identical to the staged sequence below, 14 launches)
            _lib.call("hx_sac_learn", nets, ctypes.byref(batch), hyper, ctypes.byref(pending[0]) if pending is not None else None,
                      polyak_first, self.learning_steps, self.target_entropy, st)
            return
        if self.world == 1 and not getattr(self, "separate_critic_adam", False):
            # one GPU: the critics' optimizer steps ride in the weight-gradient launch (bit-identical to the two calls below, one launch less)
            _lib.call("hx_sac_critic_step", nets, ctypes.byref(batch), hyper, ctypes.byref(pending[0]) if pending is not None else None,
                      polyak_first, self.learning_steps, st)
        else:
            if pending is not None:
                _lib.call("hx_sac_critic_grads_sampled", nets, ctypes.byref(batch), hyper, ctypes.byref(pending[0]), polyak_first, st)
            else:
                _lib.call("hx_sac_critic_grads", nets, ctypes.byref(batch), hyper, polyak_first, st)
            self._allreduce(self.grad_critic)
            _lib.call("hx_sac_adam", nets, hyper, 0, self.learning_steps, gs, self.target_entropy, st)
        _lib.call("hx_sac_policy_grads", nets, ctypes.byref(batch), hyper, st)
        if self.world > 1:  # mean entropy and the policy-loss terms are per-shard means: average them with the gradients
            self._allreduce(self.grad_policy)
            self._allreduce(self.losses)
            self.losses.mul_(gs)
        _lib.call("hx_sac_adam", nets, hyper, 1, self.learning_steps, gs, self.target_entropy, st)

    def step_learn(self, env, expert=None, n_main=None, explore=True, act_seed=0, out=None, sample_seed=0):
        """One iteration of the vector loop — act_step(env) then sample(env.replay, ..., defer=True) then learn() — in FRONT form (include/hirl4ucav.h
        hx_sac_front): the env step and the first forward launch of learn() are ONE launch, the minibatch pre-drawn by the previous call.  As in
        HirlEngine.step_learn the minibatch is drawn from the ring as it stood BEFORE this env step, without the env.n slots the step may overwrite.
        More than 8,192 envs (the persistent acting kernel), one GPU, the one-call learn(), Philox draws.  -> (actions, obs, reward, done, success)."""
        replay, n, B = env.replay, env.n, self.batch
        if self.world > 1 or replay is None or n <= 8192 or getattr(self, "separate_critic_adam", False) or getattr(self, "staged_policy", False):
            raise _lib.HxError("SacEngine.step_learn: one GPU, the one-call learn(), more than 8,192 envs with a replay ring attached")
        if getattr(self, "_pending", None) is not None:
            raise _lib.HxError("step_learn draws its own minibatch: a sample(defer=True) is still pending")
        if getattr(self, "_front_tiles", None) is None:
            second = torch.zeros(B * 32 + B, dtype=torch.float32, device=self.device)
            self._front_tiles = [(self.rows, self._idx), (second[:B * 32], second[B * 32:].view(torch.int32))]
            self._front_drawn = None
        cur, nxt = self._front_tiles
        n_main = B if (n_main is None or expert is None) else int(n_main)
        self.sample_calls += 1
        self._seed = int(sample_seed)

        def draw(tiles, call):
            return E.HxSample(replay.total.data_ptr(), replay.capacity, replay.ring.data_ptr(), expert.ring.data_ptr() if expert is not None else None,
                              E.len_of(expert), None, 0, n_main, int(sample_seed), call, 0.0, tiles[1].data_ptr(), None, n)

        want = (env, env.steps_issued, replay, expert, n_main, int(sample_seed), self.sample_calls, n)
        if self._front_drawn != want:  # no tile in waiting for THIS draw: draw now, as a launch of its own
            _lib.call("hx_sample_batch_guarded", replay.total.data_ptr(), replay.capacity, replay.ring.data_ptr(), expert.ring.data_ptr() if expert is not None else None,
                      E.len_of(expert), None, 0, B, n_main, 1, int(sample_seed), self.sample_calls, 0.0, cur[1].data_ptr(), None, self._noise.data_ptr(),
                      cur[0].data_ptr(), None, n, _lib.stream_ptr())
        if out is None:
            out = torch.empty((n, 4), dtype=torch.float32, device=self.device)
        self.act_calls += 1
        self.learning_steps += 1
        batch = HxSacBatch(cur[0].data_ptr(), B, None, None, int(sample_seed), self.learning_steps)
        nets, hyper, st = ctypes.byref(self.nets), ctypes.byref(self.hyper), _lib.stream_ptr()
        x9 = self._x9_for(n)
        _lib.call("hx_sac_front", self.policy.data_ptr(), self.w2_x9.data_ptr() if x9 else None, self.w2_f32i.data_ptr(), env.state.data_ptr(), n, env.pitch,
                  env.obs.data_ptr(), out.data_ptr(), 2 if explore else 0, None, int(act_seed), int(env.env_id0), self.act_calls, env.reward.data_ptr(),
                  env.done.data_ptr(), env.success.data_ptr(), ctypes.byref(env._opts), nets, ctypes.byref(batch), st)
        env.steps_issued += 1
        nxt_draw = draw(nxt, self.sample_calls + 1)
        polyak_first = int(self.learning_steps % self.interval == 0)
        _lib.call("hx_sac_learn_back", nets, ctypes.byref(batch), hyper, polyak_first, self.learning_steps, self.target_entropy, ctypes.byref(nxt_draw),
                  nxt[0].data_ptr(), st)
        self._front_drawn = (env, env.steps_issued, replay, expert, n_main, int(sample_seed), self.sample_calls + 1, n)
        self._front_tiles = [nxt, cur]
        self.rows, self._idx = cur
        return out, env.obs, env.reward, env.done, env.success

    def losses_host(self):
        """(q1_loss, q2_loss, policy_loss, entropy_loss, mean entropy, alpha)"""
        v = self.losses.tolist()
        return v[0], v[1], v[2], v[3], v[4], v[5]
