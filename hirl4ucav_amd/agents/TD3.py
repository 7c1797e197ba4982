"""Drop-in counterpart of agents.TD3 (reference hirl/agents/TD3.py:148-271): LeakyReLU(0.01) networks, no BC term,
learn() -> (critic_loss, actor_loss).  Same kernels as HIRL with slope = 0.01 and the BC branch off."""
import numpy as np
import torch

from ..utils.buffer import device
from .HIRL import Agent as _HirlAgent


class Agent(_HirlAgent):
    _slope, _use_bc = 0.01, False

    def __init__(self, actorLR, criticLR, stateDim, actionDim, full1Dim, full2Dim, tau, gamma, bufferSize, batchSize, layerNorm, name):
        super().__init__(actorLR, criticLR, stateDim, actionDim, full1Dim, full2Dim, tau, gamma, bufferSize, batchSize, layerNorm, name)

    def learn(self):  # TD3.py:201-260
        idx = self.buffer.sample_indices(self.batchSize)
        noise = torch.normal(mean=torch.zeros(self.actionDim), std=torch.ones(self.actionDim) * self.TD3LearningNoise)
        was_actor_call = self.eng.actor_trainable
        self.eng.assemble(self.buffer.ring, torch.as_tensor(idx, dtype=torch.int32, device=device))
        self.eng.learn(noise=noise.to(device))
        got = self.eng.losses_host()
        if was_actor_call:
            self._last = got
        else:
            self._last = (got[0],) + tuple(self._last[1:])
        return np.float32(self._last[0]), np.float32(self._last[1])
