"""HirlEngine — device-resident state of the HIRL (TD3+BC) / TD3 agent and the host-side sequencing of the update
stages exported by libhx_mi355.so (include/hirl4ucav.h).

Mirrors what hirl.agents.HIRL.Agent keeps as attributes (HIRL.py:149-190): five networks, two Adam optimisers,
`actorTrainable`, `update_count`, hyper-parameters — but every tensor is one flat fp32 buffer on the GPU and every
step only enqueues kernels on the current stream.  Sharded (one process per GPU): gradients are summed with ONE
all-reduce per phase over the flat gradient buffer (RCCL via torch.distributed), the soft-weight count with a 4-byte
all-reduce before the actor gradients are formed.
"""
import ctypes
import os

import numpy as np
import torch

from .. import _lib

H1, H2 = 256, 512
_vp, _i32, _f32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float


class HxBatch(ctypes.Structure):
    _fields_ = [("rows", _vp), ("bc_rows", _vp), ("batch", _i32), ("noise", _vp)]


class HxSample(ctypes.Structure):
    """the minibatch draw as a description (include/hirl4ucav.h HxSample): launch A of the update draws and gathers"""
    _fields_ = [("total", _vp), ("cap", ctypes.c_int64), ("ring", _vp), ("expert_ring", _vp), ("expert_len", ctypes.c_int64),
                ("bc_table", _vp), ("bc_len", ctypes.c_int64), ("n_main", _i32), ("seed", ctypes.c_uint64), ("call", ctypes.c_uint32),
                ("sigma", _f32), ("idx", _vp), ("idx_bc", _vp), ("guard", ctypes.c_uint32)]


class HxFront(ctypes.Structure):
    """in-launch hand-off state of the front launch (include/hirl4ucav.h HxFront)"""
    _fields_ = [("flags", _vp), ("status", _vp), ("epoch", ctypes.c_uint32), ("with_c", ctypes.c_uint32)]


class HxNets(ctypes.Structure):
    _fields_ = [(k, _vp) for k in ("actor", "critic", "target_actor", "target_critic", "bc_actor", "grad_actor", "grad_critic",
                                   "m_actor", "v_actor", "m_critic", "v_critic", "losses", "soft_count", "wstate", "ws", "actor_w2_bf16", "actor_w2_f32i",
                                   "w2_bf16_all", "xchg_status", "actor_w2_x9")]


class HxHyper(ctypes.Structure):
    _fields_ = [(k, _f32) for k in ("gamma", "tau", "lr_actor", "lr_critic", "slope", "noise_clamp", "loss_lambda")] + [("use_bc", _i32), ("no_layernorm", _i32)]


_P = ctypes.POINTER
_lib.register("hx_actor_act", [_vp, _vp, ctypes.c_int64, _vp, _i32, _vp, _f32, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _f32, _vp, _vp])
_lib.register("hx_actor_act_step", [_vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _i32, _vp, _f32, ctypes.c_uint64, ctypes.c_uint32,
                                     ctypes.c_uint32, _f32, _vp, _vp, _vp, _P(_lib.HxStepOpts), _vp])
_lib.register("hx_pack_w2_bf16", [_vp, _i32, _vp, _vp])
_lib.register("hx_pack_w2_f32i", [_vp, _i32, _vp, _vp])
_lib.register("hx_pack_update_images", [_P(HxNets), _vp])
_lib.register("hx_pack_w2_x9", [_vp, _i32, _vp, _vp])
_lib.register("hx_actor_act_x9", [_vp, _vp, _vp, ctypes.c_int64, _vp, _i32, _vp, _f32, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _f32, _vp])
_lib.register("hx_actor_act_step_x9", [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _i32, _vp, _f32, ctypes.c_uint64, ctypes.c_uint32,
                                        ctypes.c_uint32, _f32, _vp, _vp, _vp, _P(_lib.HxStepOpts), _vp])
_lib.register("hx_actor_act_f32i", [_vp, _vp, _vp, ctypes.c_int64, _vp, _i32, _vp, _f32, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _f32, _vp])
_lib.register("hx_actor_act_step_f32i", [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _i32, _vp, _f32, ctypes.c_uint64, ctypes.c_uint32,
                                          ctypes.c_uint32, _f32, _vp, _vp, _vp, _P(_lib.HxStepOpts), _vp])
_lib.register("hx_actor_act_bf16", [_vp, _vp, _vp, ctypes.c_int64, _vp, _i32, _vp, _f32, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _f32, _vp])
_lib.register("hx_actor_act_step_bf16", [_vp, _vp, _vp, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _i32, _vp, _f32, ctypes.c_uint64, ctypes.c_uint32,
                                          ctypes.c_uint32, _f32, _vp, _vp, _vp, _P(_lib.HxStepOpts), _vp])
_lib.register("hx_hirl_critic_grads", [_P(HxNets), _P(HxBatch), _P(HxHyper), _i32, _vp])
_lib.register("hx_hirl_actor_backward", [_P(HxNets), _P(HxBatch), _P(HxHyper), _i32, _i32, _vp])
_lib.register("hx_hirl_actor_wgrad", [_P(HxNets), _P(HxHyper), _i32, _i32, _i32, _f32, _f32, _vp])
_lib.register("hx_adam", [_P(HxNets), _P(HxHyper), _i32, _i32, _f32, _i32, _f32, _f32, _i32, _vp])
_lib.register("hx_polyak", [_P(HxNets), _P(HxHyper), _vp])
_lib.register("hx_hirl_actor_wgrad_split", [_P(HxNets), _P(HxHyper), _i32, _vp, _vp])
_lib.register("hx_adam_mixed", [_P(HxNets), _P(HxHyper), _i32, _i32, _f32, _i32, _f32, _f32, _i32, _vp, _vp])
from .exchange import OneShotExchange, RcclDirect, _DeviceWords, negotiate_rccl_direct  # noqa: E402,F401  (the transports of the sharded update)


_lib.register("hx_bc_train_actor", [_P(HxNets), _P(HxBatch), _P(HxHyper), _i32, _vp])
_lib.register("hx_hirl_learn", [_P(HxNets), _P(HxBatch), _P(HxHyper), _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp])
_lib.register("hx_hirl_learn_sampled", [_P(HxNets), _P(HxBatch), _P(HxHyper), _P(HxSample), _i32, _i32, _i32, _i32, _i32, _f32, _f32, _vp])
_lib.register("hx_hirl_front", [_vp, ctypes.c_int64, ctypes.c_int64, _vp, _vp, _i32, _vp, _f32, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32, _vp, _vp, _vp,
                                 _P(_lib.HxStepOpts), _P(HxNets), _P(HxBatch), _P(HxHyper), _i32, _i32, _P(HxFront), _vp])
_lib.register("hx_hirl_learn_back", [_P(HxNets), _P(HxBatch), _P(HxHyper), _i32, _i32, _i32, _i32, _i32, _f32, _f32, _P(HxSample), _P(HxBatch), _i32, _vp])
_lib.register("hx_hirl_critic_grads_back", [_P(HxNets), _P(HxBatch), _P(HxHyper), _P(HxSample), _P(HxBatch), _i32, _vp])
_lib.register("hx_sample_batch_guarded", [_vp, ctypes.c_int64, _vp, _vp, ctypes.c_int64, _vp, ctypes.c_int64, _i32, _i32, _i32, ctypes.c_uint64,
                                           ctypes.c_uint32, _f32, _vp, _vp, _vp, _vp, _vp, ctypes.c_uint32, _vp])
_lib.register("hx_hirl_critic_grads_sampled", [_P(HxNets), _P(HxBatch), _P(HxHyper), _P(HxSample), _i32, _vp])
_lib.register("hx_sample_batch", [_vp, ctypes.c_int64, _vp, _vp, ctypes.c_int64, _vp, ctypes.c_int64, _i32, _i32, _i32, ctypes.c_uint64,
                                   ctypes.c_uint32, _f32, _vp, _vp, _vp, _vp, _vp, _vp])

# ---- flat layout <-> reference state_dict keys (hirl/agents/HIRL.py:19-146) -------------------------------------


def _block_layout(in_dim, out_dim, names):
    """names = (full_a, ln_a, full_b, ln_b, final) -> [(key, offset, shape)] and the exact block size"""
    fa, la, fb, lb, fin = names
    shapes = [(fa + ".weight", (H1, in_dim)), (fa + ".bias", (H1,)), (la + ".weight", (H1,)), (la + ".bias", (H1,)),
              (fb + ".weight", (H2, H1)), (fb + ".bias", (H2,)), (lb + ".weight", (H2,)), (lb + ".bias", (H2,)),
              (fin + ".weight", (out_dim, H2)), (fin + ".bias", (out_dim,))]
    out, off = [], 0
    for k, shp in shapes:
        out.append((k, off, shp))
        off += int(np.prod(shp))
    return out, off


ACTOR_LAYOUT, ACTOR_SIZE = _block_layout(13, 4, ("full1", "layernorm1", "full2", "layernorm2", "final"))
_q1, _qsize = _block_layout(17, 1, ("full1", "layernorm1", "full2", "layernorm2", "final1"))
_q2, _ = _block_layout(17, 1, ("full3", "layernorm3", "full4", "layernorm4", "final2"))
Q_PADDED = (_qsize + 3) & ~3
CRITIC_LAYOUT = _q1 + [(k, off + Q_PADDED, shp) for k, off, shp in _q2]
CRITIC_SIZE = 2 * Q_PADDED


def pack(params, layout, size, device):
    flat = torch.zeros(size, dtype=torch.float32, device=device)
    for k, off, shp in layout:
        v = torch.as_tensor(np.asarray(params[k].detach().cpu() if torch.is_tensor(params[k]) else params[k]), dtype=torch.float32)
        assert tuple(v.shape) == tuple(shp), (k, v.shape, shp)
        flat[off:off + v.numel()] = v.reshape(-1).to(device)
    return flat


def unpack(flat, layout):
    return {k: flat[off:off + int(np.prod(shp))].reshape(shp) for k, off, shp in layout}


def force_plain_layernorm(flat, layout):
    """LayerNorm slots of a flat network buffer -> (1, 0).  A layerNorm=False agent never uses or trains the modules (they still exist in the
    reference's state_dict, HIRL.py:28,33,114,119); the kernels' "no LayerNorm" form multiplies by the slot's weight and adds its bias."""
    for k, off, shp in layout:
        if k.startswith("layernorm"):
            flat[off:off + int(np.prod(shp))] = 1.0 if k.endswith(".weight") else 0.0


def len_of(replay):
    """live rows of a DeviceReplay used as a fixed table (expert ring): its capacity bound, no host sync"""
    return 0 if replay is None else int(getattr(replay, "fixed_len", replay.capacity))


class HirlEngine:
    def __init__(self, batch=128, lr_actor=1e-3, lr_critic=1e-3, tau=0.005, gamma=0.99, slope=0.0, use_bc=True,
                 device="cuda", group=None, layer_norm=True):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HxError("HirlEngine runs on the GPU only (no CPU path in the product)")
        L = _lib.load()
        for i, cls in ((0, _lib.HxStepOpts), (1, HxNets), (2, HxHyper), (3, HxBatch), (4, HxSample)):
            _lib.check_struct(i, cls)  # this binding's structs against the loaded library's (include/hirl4ucav.h hx_abi_sizes)
        assert L.hx_actor_param_count() == ACTOR_SIZE and L.hx_critic_param_count() == CRITIC_SIZE
        L.hx_hirl_workspace_floats.restype = ctypes.c_int64
        self.batch = int(batch)
        # ONE arena for everything the update kernels touch (5 nets, gradients, Adam moments, workspace, minibatch tiles):
        # a single contiguous mapping keeps the number of distinct pages / translations a short kernel has to fetch small
        ws_floats = int(L.hx_hirl_workspace_floats(self.batch))
        sizes = [ACTOR_SIZE] * 3 + [CRITIC_SIZE] * 2 + [CRITIC_SIZE + ACTOR_SIZE] + [ACTOR_SIZE] * 2 + [CRITIC_SIZE] * 2 + \
                [64, 64, 64, ws_floats, self.batch * 32, self.batch * 32, self.batch, self.batch, 64]
        offs, tot = [], 0
        for n in sizes:
            offs.append(tot)
            tot += (n + 63) & ~63  # 256-B aligned carve-outs
        self.arena = torch.zeros(tot, dtype=torch.float32, device=self.device)
        carve = [self.arena[o:o + n] for o, n in zip(offs, sizes)]
        (self.actor, self.target_actor, self.bc_actor, self.critic, self.target_critic, self.grad, self.m_actor, self.v_actor,
         self.m_critic, self.v_critic, self.losses, sc, self.wstate, self.ws, self.rows, self.bc_rows, ix, ixb, self._noise) = carve
        self.grad_critic, self.grad_actor = self.grad[:CRITIC_SIZE], self.grad[CRITIC_SIZE:]
        self.soft_count, self._idx, self._idx_bc = sc.view(torch.int32), ix.view(torch.int32), ixb.view(torch.int32)
        self.losses, self.wstate, self._noise = self.losses[:8], self.wstate[:1], self._noise[:4]
        self.soft_count = self.soft_count[:1]
        self.sample_calls = 0
        self._pending = None  # a draw sample(defer=True) recorded for the next learn()
        # step_learn: hand-off words (flags int32[64], status int32[1]) + the epoch counter; two sets of minibatch tiles (the rest of learn() k still
        # reads set k while its first launch fills set k + 1 for the next front launch) and what the set in waiting was drawn for
        # launch C inside the front launch (front_c_for()).  [r5] OFF by default: with the six-term acting format the one size class where it paid (8,192 envs
        # fp32: 67.2 -> 62.6 us per step in round 4) reads 62.1 -> 61.4 us — 1 % does not earn a third in-launch dependency (profiles/r05_front_c_8192.txt).  "auto"
        # turns round 4's rule back on, True / False force it; HX_FRONT_C overrides both ways.  One process per GPU only: with launch C riding the waiting workgroups
        # outnumber the CUs, and beside two more processes on the same GPU its waits ran into their bound (profiles/r05_soak_front_shared_gpu.txt)
        self.front_c = False
        self._front = None
        self._front_enqueued = False
        self.needs_reload = False  # set when a call failed after its front launch was enqueued (step_learn); cleared by a snapshot load
        self._front_epoch = 0
        self._front_c_epoch = 0  # front launches that carried launch C (their counters advance only then)
        self._front_tiles = None
        self._front_drawn = None
        self.nets = HxNets(*(t.data_ptr() for t in (self.actor, self.critic, self.target_actor, self.target_critic, self.bc_actor,
                                                     self.grad_actor, self.grad_critic, self.m_actor, self.v_actor, self.m_critic,
                                                     self.v_critic, self.losses, self.soft_count, self.wstate, self.ws)), None, None, None, None, None)
        self.act_dtype, self.w2_bf16, self.w2_x9, self._x9_live = "f32", None, None, False
        # act_dtype "f32" and at least this many rows: the 256 -> 512 product of the ACTING kernel runs through the exact three-way bf16 split of both operands on the
        # bf16 matrix cores (hx_actor_act_x9: fp32 operands, every partial product exact, fp32 accumulation — an fp32 result up to summation
        # order, 1e-5 parity like the fp32-MFMA path) instead of fp32 MFMA: 121 against 163 us at 65,536 rows, 34 against 48 at 16,384
        # (tools/ubench/actp_time.py).  None: never (fp32 MFMA at every size).  The three images are built at the first such call and kept
        # current by the actor's Adam steps from then on.
        # [r5] from 4,096 rows on (round 4: 16,384): in the reference-order loop at 4,096 envs 68.2 -> 66.8 us per step (tools/ubench/nofront_levers.sh)
        self.x9_rows = 4096
        # ... and in the FRONT launch (step_learn) at every size: there the acting workgroups take 32 rows each on half of the CUs and are the launch's
        # longest — 21.7 against 26.8 us for 4,096 envs (tools/ubench/x9_32row.sh), 52.5 against 56.9 us per step.  False: fp32 MFMA there too.
        self.front_x9 = True
        self.update_dtype, self.images = "f32", None
        # fp32 image of the actor's W2 in the acting kernel's operand order (hx_pack_w2_f32i): kept current by the actor's Adam steps
        self.w2_f32i = torch.zeros(H2 * H1, dtype=torch.float32, device=self.device)
        self.nets.actor_w2_f32i = self.w2_f32i.data_ptr()
        # layer_norm = False: the reference's `layerNorm=False` networks (HIRL.py:70-80,92-97,135-138): both LayerNorms skipped; the modules
        # still exist in its state_dict (HIRL.py:28,33,114,119), untrained at (1, 0) — load_params keeps the flat buffers' slots there
        self.layer_norm = bool(layer_norm)
        self._mode_bits = 0 if self.layer_norm else 16  # hx_actor_act*: noise_mode + 16
        self.hyper = HxHyper(gamma, tau, lr_actor, lr_critic, slope, 0.5, 10000.0, int(use_bc), 0 if self.layer_norm else 1)  # HIRL.py:162,182
        self.use_bc, self.slope = bool(use_bc), float(slope)
        self.actor_trainable, self.update_count = True, 0   # HIRL.py:157,166
        self.critic_step, self.actor_step = 0, 0
        self.target_update_freq = 3                          # HIRL.py:183
        self.group = group
        self.world = torch.distributed.get_world_size(group) if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
        self.act_calls = 0
        self.staged = self.world > 1  # stage-by-stage path with the gradient exchanges; the one-call fused path otherwise
        # sharded runs: ONE message for the whole actor phase ([dL_rl | dL_bc | soft count], combined after the exchange)
        self.actor_msg = torch.zeros(int(L.hx_actor_message_floats()), dtype=torch.float32, device=self.device) if self.world > 1 else None
        # True at world == 1: run EXACTLY the launch sequence of a sharded rank (split actor message, hx_adam_mixed, both exchange calls
        # — through torch.distributed when a process group exists) so that its cost can be measured on one GPU (bench.py --staged)
        self.sharded_sequence = False
        self.exchange_name, self.xchg, self.rccl = "rccl", None, None

    # ---- parameters ------------------------------------------------------------------------------------------
    def load_params(self, actor, critic, bc_actor=None, hard_update_targets=True):
        self.actor.copy_(pack(actor, ACTOR_LAYOUT, ACTOR_SIZE, self.device))
        self.critic.copy_(pack(critic, CRITIC_LAYOUT, CRITIC_SIZE, self.device))
        if bc_actor is not None:
            self.bc_actor.copy_(pack(bc_actor, ACTOR_LAYOUT, ACTOR_SIZE, self.device))
        if not self.layer_norm:  # the kernels' "no LayerNorm" form multiplies by the slot's weight and adds its bias: keep them at (1, 0)
            for flat, layout in ((self.actor, ACTOR_LAYOUT), (self.critic, CRITIC_LAYOUT), (self.bc_actor, ACTOR_LAYOUT)):
                force_plain_layernorm(flat, layout)
        if hard_update_targets:  # hard_update, HIRL.py:15-17,172,176
            self.target_actor.copy_(self.actor)
            self.target_critic.copy_(self.critic)
        self.refresh_bf16()

    def set_act_dtype(self, dtype):
        """"f32": policy inference on fp32 MFMA (parity 1e-5).  "bf16": its 256 -> 512 layer on bf16 MFMA from a bf16 image of W2 that
        every actor Adam step keeps current (BASELINE.json configs[4]); what learn() computes in is set_update_dtype's business.
        "f32x9": fp32 policy inference with the 256 -> 512 product through the EXACT three-way bf16 split of both operands on the bf16 matrix cores (six of
        the nine partial products — the three below fp32 resolution are not formed; hx_actor_act_x9:
        fp32 operands, every partial product exact, fp32 accumulation — fp32 results up to summation order); fp32 update only."""
        if dtype not in ("f32", "bf16", "f32x9"):
            raise ValueError(dtype)
        if dtype == "f32x9" and self.update_dtype != "f32":
            raise ValueError("act dtype f32x9 goes with the fp32 update path")
        self.act_dtype = dtype
        self._bind_images()
        self.refresh_bf16()

    def set_update_dtype(self, dtype):
        """"f32": learn() on fp32 MFMA (parity 1e-5 vs the reference).  "bf16": the three products of every network's 256 <-> 512 layer
        (forward, input gradient, weight gradient) on bf16 MFMA with fp32 accumulation, fp32 master weights / Adam / LayerNorm / heads
        (include/hirl4ucav.h "bf16 update path"; BASELINE.json configs[4] "bf16 actor/critic")."""
        if dtype not in ("f32", "bf16"):
            raise ValueError(dtype)
        if dtype == "bf16" and self.act_dtype == "f32x9":
            raise ValueError("act dtype f32x9 goes with the fp32 update path")
        self.update_dtype = dtype
        self._bind_images()
        self.refresh_bf16()

    def _bind_images(self):
        """which bf16 buffers the kernels see: in bf16 update mode ONE block holds every image (its first image is the acting kernel's);
        with a bf16 policy only, the actor's image alone; with the exact-split format the hi | mid | lo images"""
        self.nets.actor_w2_x9 = None
        self._x9_live = False  # (an "f32" policy's large-population images are rebuilt and re-bound by the next large call: _x9_for)
        if self.update_dtype == "bf16":
            if self.images is None:
                L = _lib.load()
                L.hx_bf16_images_elems.restype = ctypes.c_int64
                self.images = torch.zeros(int(L.hx_bf16_images_elems()), dtype=torch.bfloat16, device=self.device)
            self.nets.w2_bf16_all = self.images.data_ptr()
            self.w2_bf16 = self.images[:H2 * H1]
            self.nets.actor_w2_bf16 = self.w2_bf16.data_ptr() if self.act_dtype == "bf16" else None
            return
        self.nets.w2_bf16_all = None
        if self.act_dtype == "f32x9":
            if self.w2_x9 is None:
                self.w2_x9 = torch.zeros(3 * H2 * H1, dtype=torch.bfloat16, device=self.device)  # hi | mid | lo
            self.nets.actor_w2_x9 = self.w2_x9.data_ptr()
            self.nets.actor_w2_bf16 = None
            return
        if self.act_dtype == "bf16":
            if self.w2_bf16 is None or (self.images is not None and self.w2_bf16.data_ptr() == self.images.data_ptr()):
                self.w2_bf16 = torch.zeros(H2 * H1, dtype=torch.bfloat16, device=self.device)
            self.nets.actor_w2_bf16 = self.w2_bf16.data_ptr()
        else:
            self.nets.actor_w2_bf16 = None

    def refresh_bf16(self):
        """Rebuild the kernels' images of W2 from the fp32 networks (after load_params / a checkpoint restore or any direct write to a
        network; the optimizer and Polyak steps maintain them otherwise): the actor's fp32 image always, its bf16 image with a bf16
        policy, every image of the update path in bf16 update mode."""
        _lib.call("hx_pack_w2_f32i", self.actor.data_ptr(), 13, self.w2_f32i.data_ptr(), _lib.stream_ptr())
        if self.update_dtype == "bf16":
            _lib.call("hx_pack_update_images", ctypes.byref(self.nets), _lib.stream_ptr())
        elif self.act_dtype == "bf16":
            _lib.call("hx_pack_w2_bf16", self.actor.data_ptr(), 13, self.w2_bf16.data_ptr(), _lib.stream_ptr())
        elif self.act_dtype == "f32x9" or (self.act_dtype == "f32" and self._x9_live):
            _lib.call("hx_pack_w2_x9", self.actor.data_ptr(), 13, self.w2_x9.data_ptr(), _lib.stream_ptr())

    def _x9_for(self, n, front=False):
        """True when `n` rows of the fp32 policy take the exact-split format (set_act_dtype("f32x9"), or "f32" with n >= x9_rows, or "f32" in the
        front launch with front_x9)"""
        if self.act_dtype == "f32x9":
            return True
        if self.act_dtype != "f32" or self.update_dtype != "f32":
            return False
        if not (front and self.front_x9) and (self.x9_rows is None or n < self.x9_rows):
            return False
        if not self._x9_live:  # first large call: build the images; nets.actor_w2_x9 makes every later Adam step of the actor refresh them
            if self.w2_x9 is None:
                self.w2_x9 = torch.zeros(3 * H2 * H1, dtype=torch.bfloat16, device=self.device)  # hi | mid | lo
            self.nets.actor_w2_x9 = self.w2_x9.data_ptr()
            _lib.call("hx_pack_w2_x9", self.actor.data_ptr(), 13, self.w2_x9.data_ptr(), _lib.stream_ptr())
            self._x9_live = True
        return True

    refresh_images = refresh_bf16

    def acting_format(self):
        """What decides the ACTING arithmetic of this engine beyond its dtype flags — stored in whole-run snapshots [ADVICE r5: the default `x9_rows`
        went from 16,384 to 4,096 and the split from nine to six partial products between rounds 4 and 5; a run resumed across that change continued
        under other acting arithmetic without a word]."""
        return {"act_dtype": self.act_dtype, "update_dtype": self.update_dtype, "x9_rows": self.x9_rows, "front_x9": bool(self.front_x9), "x9_terms": 6}

    def replica_checksum(self):
        """int64 sum of the bit patterns of every network and Adam moment: equal on all ranks of a sharded run, or the replicas have
        diverged (Adam and Polyak see identical inputs on every rank).  Synchronises."""
        if self.xchg is not None:
            self.xchg.check()  # a timed-out / poisoned one-shot exchange since the last check raises here
        t = torch.cat([self.actor, self.target_actor, self.critic, self.target_critic, self.m_actor, self.v_actor, self.m_critic, self.v_critic])
        return int(t.view(torch.int32).to(torch.int64).sum().item())

    def close(self):
        """Release the one-shot exchange's peer mappings / the direct RCCL communicator; raises if an exchange failed since the last check."""
        if self.rccl is not None:
            r, self.rccl = self.rccl, None
            r.close()
        if self.xchg is not None:
            x, self.xchg = self.xchg, None
            self.nets.xchg_status = None
            try:
                x.check()
            finally:
                x.close()

    def state_dicts(self):
        return {"actor": unpack(self.actor, ACTOR_LAYOUT), "critic": unpack(self.critic, CRITIC_LAYOUT),
                "targetActor": unpack(self.target_actor, ACTOR_LAYOUT), "targetCritic": unpack(self.target_critic, CRITIC_LAYOUT),
                "bc_actor": unpack(self.bc_actor, ACTOR_LAYOUT)}

    # ---- acting ----------------------------------------------------------------------------------------------
    def act(self, obs, noise=None, sigma=0.0, seed=0, row0=0, out=None, net=None):
        """clamp(actor(obs) + noise, -1, 1) for obs [N, 13] on the device (chooseAction*, HIRL.py:192-212).
        noise: None and sigma == 0 -> NoNoise; tensor [4] -> one shared draw; tensor [N, 4] -> per row; sigma > 0 and
        noise None -> Philox N(0, sigma^2) per row and component."""
        n = obs.shape[0]
        if out is None:
            out = torch.empty((n, 4), dtype=torch.float32, device=self.device)
        mode = 0
        if noise is not None:
            mode = 1 if noise.numel() == 4 else 2
        elif sigma > 0:
            mode = 3
        self.act_calls += 1
        if self.act_dtype == "bf16" and net is None:
            _lib.call("hx_actor_act_bf16", self.actor.data_ptr(), self.w2_bf16.data_ptr(), obs.data_ptr(), n, out.data_ptr(), mode | self._mode_bits, _lib.ptr(noise),
                      float(sigma), int(seed), int(row0), self.act_calls, self.slope, _lib.stream_ptr())
            return out
        if net is None and self._x9_for(n):
            _lib.call("hx_actor_act_x9", self.actor.data_ptr(), self.w2_x9.data_ptr(), obs.data_ptr(), n, out.data_ptr(), mode | self._mode_bits, _lib.ptr(noise),
                      float(sigma), int(seed), int(row0), self.act_calls, self.slope, _lib.stream_ptr())
            return out
        if net is None:  # the engine's own actor: W2 from its fp32 image (same bits as hx_actor_act, no LDS staging of W2)
            _lib.call("hx_actor_act_f32i", self.actor.data_ptr(), self.w2_f32i.data_ptr(), obs.data_ptr(), n, out.data_ptr(), mode | self._mode_bits, _lib.ptr(noise),
                      float(sigma), int(seed), int(row0), self.act_calls, self.slope, _lib.stream_ptr())
            return out
        _lib.call("hx_actor_act", net.data_ptr(), obs.data_ptr(), n, out.data_ptr(), mode | self._mode_bits,
                  _lib.ptr(noise), float(sigma), int(seed), int(row0), self.act_calls, self.slope, None,
                  _lib.stream_ptr())
        return out

    def act_step(self, env, noise=None, sigma=0.0, seed=0, out=None):
        """chooseAction for every env of `env` (a BatchedHarfangEnv) AND env.step with those actions in ONE launch
        (train_all.py:343-345): same results as act(env.obs, ...) followed by env.step(actions), bit for bit.
        -> (actions, obs, reward, done, success); env.obs holds the next observations afterwards."""
        n = env.n
        if out is None:
            out = torch.empty((n, 4), dtype=torch.float32, device=self.device)
        mode = 0
        if noise is not None:
            mode = 1 if noise.numel() == 4 else 2
        elif sigma > 0:
            mode = 3
        self.act_calls += 1
        env.steps_issued += 1
        if self.act_dtype == "bf16":
            _lib.call("hx_actor_act_step_bf16", self.actor.data_ptr(), self.w2_bf16.data_ptr(), env.state.data_ptr(), n, env.pitch, env.obs.data_ptr(),
                      out.data_ptr(), mode | self._mode_bits, _lib.ptr(noise), float(sigma), int(seed), int(env.env_id0), self.act_calls, self.slope,
                      env.reward.data_ptr(), env.done.data_ptr(), env.success.data_ptr(), ctypes.byref(env._opts), _lib.stream_ptr())
            return out, env.obs, env.reward, env.done, env.success
        if self._x9_for(n):
            _lib.call("hx_actor_act_step_x9", self.actor.data_ptr(), self.w2_x9.data_ptr(), env.state.data_ptr(), n, env.pitch, env.obs.data_ptr(),
                      out.data_ptr(), mode | self._mode_bits, _lib.ptr(noise), float(sigma), int(seed), int(env.env_id0), self.act_calls, self.slope,
                      env.reward.data_ptr(), env.done.data_ptr(), env.success.data_ptr(), ctypes.byref(env._opts), _lib.stream_ptr())
            return out, env.obs, env.reward, env.done, env.success
        _lib.call("hx_actor_act_step_f32i", self.actor.data_ptr(), self.w2_f32i.data_ptr(), env.state.data_ptr(), n, env.pitch, env.obs.data_ptr(),
                  out.data_ptr(), mode | self._mode_bits, _lib.ptr(noise), float(sigma), int(seed), int(env.env_id0), self.act_calls, self.slope,
                  env.reward.data_ptr(), env.done.data_ptr(), env.success.data_ptr(), ctypes.byref(env._opts), _lib.stream_ptr())
        return out, env.obs, env.reward, env.done, env.success

    # ---- learning --------------------------------------------------------------------------------------------
    def use_oneshot_exchange(self, timeout_ms=5000, two_stage=False, bf16=False):
        """Exchange gradients with the peer-read kernels over hipIpc mappings instead of RCCL: hx_allreduce_oneshot, or with two_stage
        hx_allreduce_twostage (reduce-scatter + all-gather; bf16: reduced slices as bf16)."""
        if self.world <= 1:
            return
        self.xchg = OneShotExchange({"critic": CRITIC_SIZE, "actor": self.actor_msg.numel()}, self.device, self.group, timeout_ms, two_stage, bf16)
        self.nets.xchg_status = self.xchg.status_ptr  # a failed exchange freezes the optimizer steps (fail-stop), check() raises
        self.exchange_name = "oneshot" if not two_stage else ("twostage-bf16" if bf16 else "twostage")

    def use_rccl_direct(self, bf16=False):
        """bf16: the two messages travel and are summed as bf16 (RcclDirect(bf16=True): half the wire bytes; replicas still bit-identical).
        Exchange gradients with ncclAllReduce enqueued by the library on the engine's stream (hx_rccl_allreduce) instead of
        torch.distributed.all_reduce: no host-side collective call inside learn().  Needs an initialised process group (for the id) and one
        GPU per rank (RCCL refuses two ranks on one device); works at world size 1 (the sharded rank's sequence, bench.py --staged).
        The communicator is checked with one all-reduce of ones before it is used, and EVERY rank takes the same decision
        (exchange.negotiate_rccl_direct: no rank is left alone in a collective, whichever step failed where): if any rank could
        not build or verify it (a second RCCL instance in the process, a launcher without device binding, ...) all ranks keep
        torch.distributed.all_reduce and say so on stderr — `exchange_name` (bench.py: `rccl_ranks.backend`) tells which transport runs."""
        dist = torch.distributed
        if not (dist.is_available() and dist.is_initialized()):
            raise _lib.HxError("use_rccl_direct needs an initialised torch.distributed process group (it carries the communicator id)")
        import sys

        def connect(uid, world, rank):
            r = RcclDirect(uid, world, rank, bf16=bf16)
            r.probe(self.device)
            return r

        r, _why = negotiate_rccl_direct(self.group, self.device if dist.get_backend(self.group) == "nccl" else "cpu", connect=connect,
                                        log=lambda m: print(m, file=sys.stderr, flush=True))
        if r is None:
            return False
        self.rccl = r
        self.exchange_name = "rccl-direct-bf16" if bf16 else "rccl-direct"
        return True

    def _allreduce(self, t, kind=None):
        """SUM over the ranks of `t` (a gradient message).  RCCL: in place.  One-shot: `t` is this rank's message buffer of `kind`, the
        sum arrives in a separate local tensor (peers are still reading `t`).  -> the tensor that holds the sum"""
        if self.world <= 1 and not (self.sharded_sequence and torch.distributed.is_available() and torch.distributed.is_initialized()):
            return t
        if self.xchg is not None and kind is not None:
            return self.xchg.allreduce(kind)
        if self.rccl is not None:
            return self.rccl.allreduce(t)
        torch.distributed.all_reduce(t, group=self.group)
        return t

    def assemble(self, ring, idx, expert_ring=None, n_main=None, bc_table=None, idx_bc=None):
        """Caller-chosen minibatch (parity tests, the N = 1 facade): gather ring[idx[r]] / expert_ring[idx[r]] and
        bc_table[idx_bc[r]] into the compact tiles the update stages read."""
        B = self.batch
        n_main = B if n_main is None else int(n_main)
        _lib.call("hx_sample_batch", None, 0, ring.data_ptr(), _lib.ptr(expert_ring), 0, _lib.ptr(bc_table), 0, B, n_main, 0, 0, 0, 0.0,
                  idx.data_ptr(), _lib.ptr(idx_bc), None, self.rows.data_ptr(), self.bc_rows.data_ptr() if bc_table is not None else None,
                  _lib.stream_ptr())

    def sample(self, replay, expert=None, bc_table=None, n_main=None, seed=0, sigma=0.2, defer=False):
        """Draw AND gather the next minibatch on the device (no host sync): UniformMemory.sample (buffer.py:38-48), the
        buffer/expert mix and np.random.choice of HIRL.py:223-251, torch.normal(0, 0.2) of HIRL.py:265.
        defer=True: nothing is launched now — the next learn() draws and gathers inside its first launch (hx_hirl_learn_sampled:
        the same indices, noise, tiles and update bit for bit, one launch less); the returned tensors are filled by that learn()."""
        B = self.batch
        self.sample_calls += 1
        if defer:
            # (the tensors are referenced by the record: they must outlive the launch)
            self._pending = (HxSample(replay.total.data_ptr(), replay.capacity, replay.ring.data_ptr(),
                                      expert.ring.data_ptr() if expert is not None else None, len_of(expert), _lib.ptr(bc_table),
                                      bc_table.shape[0] if bc_table is not None else 0, B if n_main is None else int(n_main), int(seed),
                                      self.sample_calls, float(sigma), self._idx.data_ptr(),
                                      self._idx_bc.data_ptr() if bc_table is not None else None), replay, expert, bc_table)
            return self._idx, self._idx_bc, self._noise
        self._pending = None
        _lib.call("hx_sample_batch", replay.total.data_ptr(), replay.capacity, replay.ring.data_ptr(),
                  expert.ring.data_ptr() if expert is not None else None, len_of(expert), _lib.ptr(bc_table),
                  bc_table.shape[0] if bc_table is not None else 0, B, B if n_main is None else int(n_main), 1, int(seed),
                  self.sample_calls, float(sigma), self._idx.data_ptr(), self._idx_bc.data_ptr() if bc_table is not None else None,
                  self._noise.data_ptr(), self.rows.data_ptr(), self.bc_rows.data_ptr() if bc_table is not None else None, _lib.stream_ptr())
        return self._idx, self._idx_bc, self._noise

    def learn(self, noise=None, bc_weight_now=0.0, bc_warm_up_weight=0.0, before_exchange=None):
        """One Agent.learn (HIRL.py:221-334 / TD3.py:201-260) on the minibatch last assembled by sample() / assemble().
        noise: the (4,) target-smoothing draw (default: the one sample() drew).  bc_weight_now: 100 = estimate the soft
        weight now (HIRL.py:299), None = keep the stored device value, else the given weight.  Enqueues only; read the
        results with losses_host().  before_exchange: called once, right before the first gradient all-reduce of the sharded path
        (utils/pipeline.py releases its side stream there)."""
        B = self.batch
        if self.needs_reload:
            raise _lib.HxError("this engine failed after a front launch had been enqueued: load a snapshot before learn()")
        st = _lib.stream_ptr()
        pending, self._pending = self._pending, None
        pipe = getattr(before_exchange, "__self__", None) if before_exchange is not None else None
        if pending is not None and pipe is not None and (getattr(pipe, "_armed", None) is not None or getattr(pipe, "issued", False)):
            # a deferred draw reads *total and the ring INSIDE learn()'s first launch; an env step released beside that launch inserts into the
            # ring and moves *total under it (workgroups would disagree about the live length).  Draw first, then overlap.
            raise _lib.HxError("sample(defer=True) cannot be combined with an armed VectorStepPipeline: use sample(defer=False) when the "
                               "next env step may run beside learn()")
        if pending is not None and noise is not None:
            raise _lib.HxError("learn(noise=...) after sample(defer=True): the deferred draw produces the smoothing noise itself")
        noise = self._noise if noise is None else noise
        batch = HxBatch(self.rows.data_ptr(), self.bc_rows.data_ptr() if self.use_bc else None, B, noise.data_ptr())
        smp = ctypes.byref(pending[0]) if pending is not None else None
        nets, hyper = ctypes.byref(self.nets), ctypes.byref(self.hyper)
        if bc_weight_now is None:
            w_kind, w_given = 2, 0.0
        elif bc_weight_now == 100:
            w_kind, w_given = 1, 0.0
        else:
            w_kind, w_given = 0, float(bc_weight_now)
        actor_phase = self.actor_trainable  # HIRL.py:291
        self.critic_step += 1
        if actor_phase:
            self.actor_step += 1
            self.update_count += 1
        do_polyak = actor_phase and self.update_count % self.target_update_freq == 0  # HIRL.py:327-330
        if not self.staged:
            _lib.call("hx_hirl_learn_sampled", nets, ctypes.byref(batch), hyper, smp, self.critic_step, int(actor_phase), self.actor_step,
                      int(do_polyak), w_kind, w_given, float(bc_warm_up_weight), st)
        else:
            actor_fwd = (2 if w_kind == 1 else 1) if actor_phase else 0
            if smp is not None:
                grads = lambda: _lib.call("hx_hirl_critic_grads_sampled", nets, ctypes.byref(batch), hyper, smp, actor_fwd, st)  # noqa: E731
            else:
                grads = lambda: _lib.call("hx_hirl_critic_grads", nets, ctypes.byref(batch), hyper, actor_fwd, st)  # noqa: E731
            self._learn_staged(grads, batch, actor_phase, do_polyak, w_kind, w_given, bc_warm_up_weight, before_exchange)
        self.actor_trainable = not self.actor_trainable  # HIRL.py:332

    def _learn_staged(self, critic_grads, batch, actor_phase, do_polyak, w_kind, w_given, bc_warm_up_weight, before_exchange=None):
        """sharded: the stages of learn() with the exchanges of SURVEY.md 8e in between — ONE message per phase.  critic_grads() issues the launches
        that leave the critics' gradient behind (all of them, or — after a front launch — the rest of them)."""
        B, st = self.batch, _lib.stream_ptr()
        nets, hyper = ctypes.byref(self.nets), ctypes.byref(self.hyper)
        gs = 1.0 / self.world
        own_critic = self.grad_critic.data_ptr()
        if self.xchg is not None:  # peers read this rank's gradient straight out of its message buffer: wgrad writes there
            self.nets.grad_critic = self.xchg.write_buffer("critic").data_ptr()
        critic_grads()
        if before_exchange is not None:
            before_exchange()
        summed = self._allreduce(self.grad_critic, "critic")
        self.nets.grad_critic = summed.data_ptr()
        pk = 16 if do_polyak else 0  # + 16: soft_update of the target in the same launch (nothing reads it in between)
        _lib.call("hx_adam", nets, hyper, 0 | pk, self.critic_step, gs, 0, 0.0, 0.0, B, st)
        self.nets.grad_critic = own_critic
        if actor_phase:
            _lib.call("hx_hirl_actor_backward", nets, ctypes.byref(batch), hyper, int(w_kind == 1), 1, st)
            if self.world > 1 or self.sharded_sequence:   # [dL_rl | dL_bc | count] in one message, w formed from the global count after the exchange
                if self.actor_msg is None:
                    self.actor_msg = torch.zeros(int(_lib.load().hx_actor_message_floats()), dtype=torch.float32, device=self.device)
                msg = self.xchg.write_buffer("actor") if self.xchg is not None else self.actor_msg
                _lib.call("hx_hirl_actor_wgrad_split", nets, hyper, B, msg.data_ptr(), st)
                summed = self._allreduce(msg, "actor")
                _lib.call("hx_adam_mixed", nets, hyper, int(do_polyak), self.actor_step, gs, w_kind, w_given, float(bc_warm_up_weight),
                          B * self.world, summed.data_ptr(), st)
            else:                # one rank running the staged sequence (tests): the weight is known locally, bit-identical to the one-call path
                _lib.call("hx_hirl_actor_wgrad", nets, hyper, B, B, w_kind, w_given, float(bc_warm_up_weight), st)
                _lib.call("hx_adam", nets, hyper, 1 | pk, self.actor_step, gs, w_kind, w_given, float(bc_warm_up_weight), B, st)

    def step_learn(self, env, expert=None, bc_table=None, n_main=None, act_noise=None, act_sigma=0.0, act_seed=0, out=None, sample_seed=0,
                   smooth_sigma=0.2, bc_weight_now=0.0, bc_warm_up_weight=0.0):
        """One iteration of the vector loop — act_step(env) then sample(env.replay, ..., defer=True) then learn() — in FRONT form
        (include/hirl4ucav.h hx_hirl_front): the env step and the first two launches of learn() are ONE launch, the acting workgroups on half of
        the CUs, the update's on the other half.  The one change of meaning: the minibatch is drawn from the ring as it stood BEFORE this env
        step, without the env.n slots the step may overwrite (uniform over every transition that is in the buffer before and after the step).
        Each call also draws the NEXT call's minibatch (inside its learn() part, after this step's inserts): a next call with the same tables,
        n_main, seed and sigma finds its tiles ready, any other draws them with a launch of its own first.
        fp32 networks, or bf16 acting + bf16 update; at most 8,192 envs per GPU, batch <= 256; one-call and sharded (staged) update paths.
        -> (actions, obs, reward, done, success) as act_step."""
        replay, n, B = env.replay, env.n, self.batch
        bf16 = self.update_dtype == "bf16"
        if replay is None or (self.act_dtype != "bf16" if bf16 else self.act_dtype not in ("f32", "f32x9")):
            raise _lib.HxError("step_learn: the front launch exists for the fp32 networks (acting format 'f32' or 'f32x9') and for the bf16 update path with "
                               "the bf16 acting format, with a replay ring attached to the env")
        if self._pending is not None:
            raise _lib.HxError("step_learn draws its own minibatch: a sample(defer=True) is still pending")
        if self._front is None:
            buf = torch.zeros(96, dtype=torch.int32, device=self.device)
            self._front = (buf[0:64], buf[64:65])
            second = torch.zeros(2 * B * 32 + 2 * B + 64, dtype=torch.float32, device=self.device)
            t2 = (second[:B * 32], second[B * 32:2 * B * 32], second[2 * B * 32:2 * B * 32 + B].view(torch.int32),
                  second[2 * B * 32 + B:2 * B * 32 + 2 * B].view(torch.int32), second[2 * B * 32 + 2 * B:2 * B * 32 + 2 * B + 4])
            self._front_tiles = [(self.rows, self.bc_rows, self._idx, self._idx_bc, self._noise), t2]
        flags, status = self._front
        cur, nxt = self._front_tiles
        n_main = B if n_main is None else int(n_main)
        # the host's counters run ahead of the library calls below; if one of them refuses BEFORE the front launch is enqueued (HX_REQUIRE: ring smaller than
        # 2n, too many row tiles, a missing image) they are put back and the hand-off words start over — a caller that catches the error must not be left
        # with a host epoch ahead of the device's.  A failure AFTER hx_hirl_front has returned (the rest of learn(), a staged exchange) is another matter
        # [ADVICE r5]: the device HAS stepped the envs, inserted into the ring and maybe applied part of the update — putting the host's counters back would
        # replay the same Philox noise and sample calls against a device that has moved on.  The counters stay advanced and the engine is marked:
        # `needs_reload` refuses every further step_learn / learn until a snapshot has been loaded (utils/checkpoint.load_engine_state clears it).
        if self.needs_reload:
            raise _lib.HxError("this engine failed after a front launch had been enqueued (host and device state may disagree): load a snapshot "
                               "(utils/checkpoint.load_run) before stepping it again")
        held = (self._front_epoch, self._front_c_epoch, self.critic_step, self.actor_step, self.update_count, self.sample_calls, self.act_calls, env.steps_issued)
        self._front_enqueued = False
        try:
            return self._step_learn(env, expert, bc_table, n_main, act_noise, act_sigma, act_seed, out, sample_seed, smooth_sigma, bc_weight_now,
                                    bc_warm_up_weight, bf16, flags, status, cur, nxt)
        except _lib.HxError:
            if self._front_enqueued:
                self.needs_reload = True
                raise
            (self._front_epoch, self._front_c_epoch, self.critic_step, self.actor_step, self.update_count, self.sample_calls, self.act_calls, env.steps_issued) = held
            self.front_reset()
            raise

    def _step_learn(self, env, expert, bc_table, n_main, act_noise, act_sigma, act_seed, out, sample_seed, smooth_sigma, bc_weight_now, bc_warm_up_weight,
                    bf16, flags, status, cur, nxt):
        replay, n, B = env.replay, env.n, self.batch
        self.sample_calls += 1

        def draw(tiles, call):
            return HxSample(replay.total.data_ptr(), replay.capacity, replay.ring.data_ptr(), expert.ring.data_ptr() if expert is not None else None,
                            len_of(expert), _lib.ptr(bc_table), bc_table.shape[0] if bc_table is not None else 0, n_main, int(sample_seed), call,
                            float(smooth_sigma), tiles[2].data_ptr(), tiles[3].data_ptr() if bc_table is not None else None, n)

        def tiles_of(t):
            return HxBatch(t[0].data_ptr(), t[1].data_ptr() if self.use_bc else None, B, t[4].data_ptr())

        want = (env, env.steps_issued, replay, expert, bc_table, n_main, int(sample_seed), float(smooth_sigma), self.sample_calls, n)
        if self._front_drawn != want:  # no tiles in waiting for THIS draw (first call, another env step since, other tables): draw now, as a launch of its own
            _lib.call("hx_sample_batch_guarded", replay.total.data_ptr(), replay.capacity, replay.ring.data_ptr(), expert.ring.data_ptr() if expert is not None else None,
                      len_of(expert), _lib.ptr(bc_table), bc_table.shape[0] if bc_table is not None else 0, B, n_main, 1, int(sample_seed), self.sample_calls,
                      float(smooth_sigma), cur[2].data_ptr(), cur[3].data_ptr() if bc_table is not None else None, cur[4].data_ptr(), cur[0].data_ptr(),
                      cur[1].data_ptr() if bc_table is not None else None, n, _lib.stream_ptr())
        if out is None:
            out = torch.empty((n, 4), dtype=torch.float32, device=self.device)
        mode = 0
        if act_noise is not None:
            mode = 1 if act_noise.numel() == 4 else 2
        elif act_sigma > 0:
            mode = 3
        self.act_calls += 1
        if self._front_epoch >= 200_000_000:  # the tiles' counters advance by up to 16 per launch: start over long before 32 bits run out (stream-ordered reset)
            flags.zero_()
            self._front_epoch = self._front_c_epoch = 0
        self._front_epoch += 1
        batch, nets, hyper, st = tiles_of(cur), ctypes.byref(self.nets), ctypes.byref(self.hyper), _lib.stream_ptr()
        if bc_weight_now is None:
            w_kind, w_given = 2, 0.0
        elif bc_weight_now == 100:
            w_kind, w_given = 1, 0.0
        else:
            w_kind, w_given = 0, float(bc_weight_now)
        actor_phase = self.actor_trainable  # HIRL.py:291
        self.critic_step += 1
        if actor_phase:
            self.actor_step += 1
            self.update_count += 1
        do_polyak = actor_phase and self.update_count % self.target_update_freq == 0  # HIRL.py:327-330
        with_c = int(self.front_c_for(n, actor_phase, w_kind, bf16))
        knob = os.environ.get("HX_FRONT_C")  # (the library's A/B knob overrides the caller both ways; forced in, launch C needs its count as well)
        carries_c = knob == "1" or (bool(with_c) and knob != "0")
        if carries_c:
            self._front_c_epoch += 1  # the counters of launch C's producers advance only in launches that carry it
        front = HxFront(flags.data_ptr(), status.data_ptr(), self._front_epoch, self._front_c_epoch if carries_c else 0)
        _lib.call("hx_hirl_front", env.state.data_ptr(), n, env.pitch, env.obs.data_ptr(), out.data_ptr(), mode | self._mode_bits | (32 if (not bf16 and self._x9_for(n, front=True)) else 0),
                  _lib.ptr(act_noise), float(act_sigma), int(act_seed), int(env.env_id0), self.act_calls, env.reward.data_ptr(), env.done.data_ptr(),
                  env.success.data_ptr(), ctypes.byref(env._opts), nets, ctypes.byref(batch), hyper, int(actor_phase), w_kind, ctypes.byref(front), st)
        self._front_enqueued = True  # from here on a failure leaves the device AHEAD of any rollback (step_learn)
        env.steps_issued += 1
        nxt_draw, nxt_tiles = draw(nxt, self.sample_calls + 1), tiles_of(nxt)
        if not self.staged:
            _lib.call("hx_hirl_learn_back", nets, ctypes.byref(batch), hyper, self.critic_step, int(actor_phase), self.actor_step, int(do_polyak), w_kind, w_given,
                      float(bc_warm_up_weight), ctypes.byref(nxt_draw), ctypes.byref(nxt_tiles), with_c, st)
        else:  # a sharded rank: the same front launch (every rank draws from its own ring), then the stages with the exchanges in between
            self._learn_staged(lambda: _lib.call("hx_hirl_critic_grads_back", nets, ctypes.byref(batch), hyper, ctypes.byref(nxt_draw), ctypes.byref(nxt_tiles), with_c, st),
                               batch, actor_phase, do_polyak, w_kind, w_given, bc_warm_up_weight)
        self._front_drawn = (env, env.steps_issued, replay, expert, bc_table, n_main, int(sample_seed), float(smooth_sigma), self.sample_calls + 1, n)
        self._front_tiles = [nxt, cur]
        self.rows, self.bc_rows, self._idx, self._idx_bc, self._noise = cur  # what this call's learn() read (the attribute names sample() / learn() use)
        self.actor_trainable = not self.actor_trainable  # HIRL.py:332
        return out, env.obs, env.reward, env.done, env.success

    @staticmethod
    def front_waiting_workgroups(batch, with_c=False):
        """How many workgroups of ONE front launch wait in-launch for others: launch B's two target-critic jobs (a 64-column workgroup per 16-row tile and
        column slice: 8 per tile) and, with launch C riding, its two TD jobs (8 column workgroups per 8 rows).  Every workgroup of the launch is a whole CU
        (1,024 threads, up to 128 VGPRs), the acting workgroups and launch A's never wait and leave within ~20 us — so while this count stays BELOW the
        number of CUs (256 on MI355X), pending producers always find a CU under ANY dispatch order that places pending workgroups on free CUs, and every
        wait ends long before its ~1 s bound.  Default shape (B = 128, launch C on its own): 128 of 256.  At B = 256 (256) or with launch C riding
        (128 + 256) the count reaches the chip, and the waits lean on the dispatch order observed on gfx950: producers (lower indices) start first
        (include/hirl4ucav.h hx_hirl_front; bounded waits + sticky status word + train_all's fallback cover that case)."""
        tiles = (int(batch) + 15) // 16
        return 2 * tiles * 8 + (2 * ((int(batch) + 7) // 8) * 8 if with_c else 0)

    def front_c_for(self, n, actor_phase, w_kind, bf16):
        """Does launch C (the critics' backward) ride in the front launch of this call too (HxFront.with_c)?  `front_c`: True / False, or "auto" (default):
        only where the acting workgroups leave the other CUs more time than the update's workgroups need — CU time bounds the front launch
        (profiles/archive/r04c_front_c_ab.txt).  That is the STREAMING acting role (fp32 in the exact-split format, 8,192 .. 16,384 envs: one 64-row pass of ~37 us
        per acting workgroup) with few enough acting workgroups: forward workgroups cost ~6 us of a CU each, launch C's 7.6 (their rows are asked for
        behind the in-launch wait).  Around 8,192 envs: 67.1 -> 62.2 us per step; everywhere else it is slower (4,096 envs 54.1 -> 56.6 us), hence off."""
        if self.front_c != "auto":
            return bool(self.front_c)
        if bf16 or not self._x9_for(n, front=True) or n < 8192 or n > 16384:
            return False
        acting = (n + 63) // 64
        # (the job count of an ACTOR call whatever this call is: on for critic-only calls alone it was measured neutral, 8,704 .. 11,264 envs)
        jobs = (3 + 2) + (1 + (0 if not self.use_bc else (2 if w_kind == 1 else 1)))
        tiles = (self.batch + 15) // 16
        need = jobs * tiles * 8 * 6.0 + 2 * ((self.batch + 7) // 8) * 8 * 7.6
        return (256 - acting) * 37.0 >= need

    def front_reset(self):
        """The hand-off words of the front launch start over (stream-ordered): counters and status to zero, the host's epochs with them, no tiles in waiting
        (the next step_learn draws its minibatch with a launch of its own)."""
        if self._front is not None:
            self._front[0].zero_()
            self._front[1].zero_()
        self._front_epoch = self._front_c_epoch = 0
        self._front_drawn = None

    def front_status(self):
        """The front launch's sticky status word (synchronises).  0: every in-launch wait was answered.  Bit 0: a launch-B workgroup gave up waiting for the
        target actor's rows of its row tile (launch A); bit 1: a launch-C workgroup gave up waiting for the jobs of A / B it reads.  In the default shape (B = 128, launch C on its own) the waiting
        workgroups are half a chip's worth and the waits end under any dispatch order (front_waiting_workgroups); where they can fill the chip (B = 256, launch C riding) the waits
        assume that the workgroups of one launch START in index order (producers have lower indices) — observed on gfx950, promised by nobody (include/hirl4ucav.h
        hx_hirl_front).  A set bit means a minibatch may have been read half-written, and every update since is suspect."""
        return 0 if self._front is None else int(self._front[1].item())

    def front_check(self):
        """Raise if a workgroup of a front launch ever gave up waiting for its producers (synchronises).  `train_all --loop reference` /
        `bench.py --no-front` run the same update without in-launch waits."""
        code = self.front_status()
        if code != 0:
            which = [w for bit, w in ((1, "launch B for the target actor's rows (launch A)"), (2, "launch C for the rows of launches A / B")) if code & bit]
            raise _lib.HxError(f"front launch: an in-launch wait timed out (status word {code}: " + "; ".join(which) + ") - the minibatches read since are "
                               "suspect; the reference-order loop (train_all --loop reference) has no in-launch waits")

    def bc_train_actor(self):
        """BC.Agent.train_actor (BC.py:160-185) on the BC minibatch last assembled: mse, backward, Adam on the actor."""
        batch = HxBatch(self.rows.data_ptr(), self.bc_rows.data_ptr(), self.batch, self._noise.data_ptr())
        self.actor_step += 1
        _lib.call("hx_bc_train_actor", ctypes.byref(self.nets), ctypes.byref(batch), ctypes.byref(self.hyper), self.actor_step, _lib.stream_ptr())

    def losses_host(self):
        """(critic_loss, actor_loss, bc_loss, rl_loss, bc_fire_loss, bc_weight) — HIRL.py:334.  Synchronises."""
        v = self.losses.tolist()
        return v[0], v[1], v[2], v[3], v[4], v[5]
