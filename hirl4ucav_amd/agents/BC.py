"""Drop-in counterpart of agents.BC (reference hirl/agents/BC.py:146-197): behaviour cloning of the LeakyReLU actor on
the expert (s, a) pairs; it produces the bc_actor checkpoints HIRL-soft consumes (train_all.py:311-312)."""
import numpy as np
import torch

from ..utils.buffer import device
from . import engine as E
from .HIRL import _check_dims, _NetView, init_actor_state_dict


class Agent:
    def __init__(self, actorLR, stateDim, actionDim, full1Dim, full2Dim, layerNorm, name, batchsize, expert_states, expert_actions):
        _check_dims(stateDim, actionDim, full1Dim, full2Dim, layerNorm)
        self.batchsize = int(batchsize)
        self.expert_states, self.expert_actions = np.asarray(expert_states), np.asarray(expert_actions)
        self.eng = E.HirlEngine(batch=self.batchsize, lr_actor=actorLR, slope=0.01, use_bc=True, device=device, layer_norm=bool(layerNorm))  # BC.py:129-135 leaky_relu
        self.eng.load_params(init_actor_state_dict(), E.unpack(self.eng.critic, E.CRITIC_LAYOUT))
        self.actor = _NetView(self.eng.actor, E.ACTOR_LAYOUT, name, on_load=self.eng.refresh_images)
        tab = np.zeros((len(self.expert_states), 32), np.float32)
        tab[:, 0:13], tab[:, 13:17] = self.expert_states, self.expert_actions
        self._bc_table = torch.from_numpy(tab).to(device)
        self.target_indices = np.where(self.expert_actions[:, 3] == 1)[0]  # BC.py:156-158
        self.expert_upsample = False

    def train_actor(self):  # BC.py:160-185
        idx = np.random.choice(self.expert_states.shape[0], self.batchsize, replace=False)
        t = torch.as_tensor(idx.astype(np.int32), device=device)
        self.eng.assemble(self._bc_table, t, bc_table=self._bc_table, idx_bc=t)
        self.eng.bc_train_actor()
        return np.float32(self.eng.losses_host()[2])

    def saveCheckpoints(self, ajan, model_name):
        self.actor.saveCheckpoint(ajan, model_name)

    def loadCheckpoints(self, ajan, model_name):
        self.actor.loadCheckpoint(ajan, model_name)

    def chooseActionNoNoise(self, state):  # BC.py:193-197
        obs = torch.as_tensor(np.asarray(state, np.float32).reshape(1, 13)).to(device)
        return self.eng.act(obs)[0].cpu().numpy()
