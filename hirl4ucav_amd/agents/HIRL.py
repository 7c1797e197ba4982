"""Drop-in counterpart of hirl.agents.HIRL (reference hirl/agents/HIRL.py): same constructor arguments, same
chooseAction* / store / learn / checkpoint methods and the same six return values from learn(), running on the HIP
kernels behind include/hirl4ucav.h.  Randomness comes from the same host generators the reference uses (`random`,
`numpy.random`, `torch.normal`), so set_seed() keeps its meaning; the fast, sync-free path is HirlEngine + the
vectorised driver (hirl4ucav_amd/train_all.py).
"""
import os

import numpy as np
import torch

from ..utils.buffer import UniformMemory, device  # noqa: F401
from . import engine as E


class _NetView:
    """Stands in for the reference's nn.Module attributes (agent.actor, agent.critic, ...): state_dict() /
    load_state_dict() / saveCheckpoint / loadCheckpoint with the reference's key names (HIRL.py:99-103,142-146)."""

    def __init__(self, flat, layout, name, on_load=None, layer_norm=True):
        self._flat, self._layout, self.name = flat, layout, name
        self.device = flat.device
        self._on_load = on_load  # the acting network: the engine's images of its W2 must follow a direct write
        self._layer_norm = layer_norm

    def state_dict(self):
        return {k: v.detach().clone().cpu() for k, v in E.unpack(self._flat, self._layout).items()}

    def load_state_dict(self, sd):
        self._flat.copy_(E.pack(sd, self._layout, self._flat.numel(), self._flat.device))
        if not self._layer_norm:
            E.force_plain_layernorm(self._flat, self._layout)
        if self._on_load is not None:
            self._on_load()

    def _path(self, ajan, model_name):
        return os.path.join(model_name, "{}".format(ajan) + self.name)  # the reference joins with a literal '\\'

    def saveCheckpoint(self, ajan, model_name):
        torch.save(self.state_dict(), self._path(ajan, model_name))

    def loadCheckpoint(self, ajan, model_name):
        p = self._path(ajan, model_name)
        if not os.path.exists(p):  # a file written by the reference on Linux is literally named 'dir\\tagName'
            p = model_name + "\\{}".format(ajan) + self.name
        self.load_state_dict(torch.load(p, map_location="cpu"))

    def eval(self):
        return self

    def train(self):
        return self


def _check_dims(stateDim, actionDim, full1Dim, full2Dim, layerNorm=True):
    """layerNorm may be True or False (both branches of HIRL.py:58-80,131-138 have kernels); the layer widths are the reference's"""
    if (stateDim, actionDim, full1Dim, full2Dim) != (13, 4, 256, 512):
        raise NotImplementedError("the HIP kernels are built for the reference's network shape 13/4/256/512 (train_all.py:190-208)")


def _init_block(in_dim, out_dim, names):
    """The reference's initialisation: kaiming_uniform_(a=0.01, fan_in, 'relu') on hidden weights, nn.Linear defaults
    elsewhere, LayerNorm (1, 0)  (HIRL.py:26-37,111-121)."""
    import torch.nn as nn

    fa, la, fb, lb, fin = names
    l1, l2, l3 = nn.Linear(in_dim, 256), nn.Linear(256, 512), nn.Linear(512, out_dim)
    nn.init.kaiming_uniform_(l1.weight, a=0.01, mode="fan_in", nonlinearity="relu")
    nn.init.kaiming_uniform_(l2.weight, a=0.01, mode="fan_in", nonlinearity="relu")
    sd = {}
    for name, lin in ((fa, l1), (fb, l2), (fin, l3)):
        sd[name + ".weight"], sd[name + ".bias"] = lin.weight.detach(), lin.bias.detach()
    for name, n in ((la, 256), (lb, 512)):
        sd[name + ".weight"], sd[name + ".bias"] = torch.ones(n), torch.zeros(n)
    return sd


def init_actor_state_dict():
    return _init_block(13, 4, ("full1", "layernorm1", "full2", "layernorm2", "final"))


def init_critic_state_dict():
    sd = _init_block(17, 1, ("full1", "layernorm1", "full2", "layernorm2", "final1"))
    sd.update(_init_block(17, 1, ("full3", "layernorm3", "full4", "layernorm4", "final2")))
    return sd


class Agent:
    _slope, _use_bc = 0.0, True

    def __init__(self, actorLR, criticLR, stateDim, actionDim, full1Dim, full2Dim, tau, gamma, bufferSize, batchSize,
                 layerNorm, name, expert_states=None, expert_actions=None, bc_weight=0.0, expert_warm_up=False):
        _check_dims(stateDim, actionDim, full1Dim, full2Dim, layerNorm)
        self.device = device
        self.tau, self.gamma, self.bufferSize, self.batchSize = tau, gamma, int(bufferSize), int(batchSize)
        self.actionDim = actionDim
        self.actionNoise, self.TD3LearningNoise, self.TD3LearningNoiseClamp = 0.1, 0.2, 0.5  # HIRL.py:160-162
        self.bc_weight, self.expert_warm_up = bc_weight, expert_warm_up
        self.layerNorm = bool(layerNorm)
        self.eng = E.HirlEngine(batch=batchSize, lr_actor=actorLR, lr_critic=criticLR, tau=tau, gamma=gamma, slope=self._slope,
                                use_bc=self._use_bc, device=device, layer_norm=self.layerNorm)
        self.eng.load_params(init_actor_state_dict(), init_critic_state_dict(), init_actor_state_dict() if self._use_bc else None)
        e = self.eng
        ln = self.layerNorm
        self.actor = _NetView(e.actor, E.ACTOR_LAYOUT, "Actor_" + name, on_load=e.refresh_images, layer_norm=ln)
        self.targetActor = _NetView(e.target_actor, E.ACTOR_LAYOUT, "TargetActor_" + name, on_load=e.refresh_images, layer_norm=ln)
        self.critic = _NetView(e.critic, E.CRITIC_LAYOUT, "Critic_" + name, on_load=e.refresh_images, layer_norm=ln)
        self.targetCritic = _NetView(e.target_critic, E.CRITIC_LAYOUT, "TargetCritic_" + name, on_load=e.refresh_images, layer_norm=ln)
        self.bc_actor = _NetView(e.bc_actor, E.ACTOR_LAYOUT, name, on_load=e.refresh_images, layer_norm=ln)
        self.buffer = UniformMemory(bufferSize, False)
        if self._use_bc:
            self.expert_states, self.expert_actions = np.asarray(expert_states), np.asarray(expert_actions)
            self.expert_buffer = UniformMemory(len(self.expert_states) + 10, False)  # HIRL.py:190
            tab = np.zeros((len(self.expert_states), 32), np.float32)
            tab[:, 0:13], tab[:, 13:17] = self.expert_states, self.expert_actions
            self._bc_table = torch.from_numpy(tab).to(device)
        self._last = (0.0, 0.0, 0.0, 0.0, 0.0, bc_weight)

    @property
    def actorTrainable(self):
        return self.eng.actor_trainable

    @property
    def update_count(self):
        return self.eng.update_count

    # ---- acting: HIRL.py:192-212 ------------------------------------------------------------------------------
    def _act(self, state, std):
        obs = torch.as_tensor(np.asarray(state, np.float32).reshape(1, 13)).to(device)
        noise = None
        if std is not None:
            noise = torch.normal(mean=torch.zeros(self.actionDim), std=torch.ones(self.actionDim) * std).to(device)
        return self.eng.act(obs, noise=noise)[0].cpu().numpy()

    def chooseAction(self, state):
        return self._act(state, self.actionNoise)

    def chooseActionSmallNoise(self, state):
        return self._act(state, self.actionNoise / 10)

    def chooseActionNoNoise(self, state):
        return self._act(state, None)

    def store(self, *args):  # HIRL.py:214-215
        self.buffer.store(*args)

    # ---- learning: HIRL.py:221-334 ----------------------------------------------------------------------------
    def learn(self, bc_weight_now, expert_num_now, bc_warm_up_weight=0):
        B = self.batchSize
        if self.expert_warm_up and expert_num_now:
            idx = self.buffer.sample_indices(B - expert_num_now) + self.expert_buffer.sample_indices(expert_num_now)
            n_main = B - expert_num_now
        else:
            idx, n_main = self.buffer.sample_indices(B), B
        idx_bc = np.random.choice(self.expert_states.shape[0], B, replace=False)  # HIRL.py:249
        noise = torch.normal(mean=torch.zeros(self.actionDim), std=torch.ones(self.actionDim) * self.TD3LearningNoise)  # :265
        was_actor_call = self.eng.actor_trainable
        self.eng.assemble(self.buffer.ring, torch.as_tensor(idx, dtype=torch.int32, device=device), expert_ring=self.expert_buffer.ring,
                          n_main=n_main, bc_table=self._bc_table, idx_bc=torch.as_tensor(idx_bc.astype(np.int32), device=device))
        self.eng.learn(noise=noise.to(device), bc_weight_now=bc_weight_now, bc_warm_up_weight=bc_warm_up_weight)
        got = self.eng.losses_host()
        if was_actor_call:
            self._last = got
            self.bc_weight = got[5]
        else:  # the last five values are stale from the previous actor call (HIRL.py:292,334)
            self._last = (got[0],) + tuple(self._last[1:])
        return tuple(np.float32(v) for v in self._last[:4]) + (self._last[4], self._last[5])

    # ---- checkpoints: HIRL.py:336-350 -------------------------------------------------------------------------
    def saveCheckpoints(self, ajan, model_name):
        for net in (self.critic, self.actor, self.targetCritic, self.targetActor):
            net.saveCheckpoint(ajan, model_name)

    def loadCheckpoints(self, ajan, model_name):
        for net in (self.critic, self.actor, self.targetCritic, self.targetActor):
            net.loadCheckpoint(ajan, model_name)

    def load_bc_actor(self, ajan, model_name):
        self.bc_actor.loadCheckpoint(ajan, model_name)
