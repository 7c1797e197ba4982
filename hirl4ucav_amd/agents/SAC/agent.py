"""Drop-in counterpart of agents.SAC.agent.SacAgent (reference hirl/agents/SAC/agent.py:58-448) for what train_sac.py uses:
constructor keywords, .memory.append / len(.memory), .batch_size, .explore / .exploit, .learn(if_expert, expert_num,
expert_data), .expert_memory, .writer, .model_dir / .plot_dir, .save_models — on the HIP kernels (SacEngine).

The networks are the Linear-ReLU stacks of the reference's un-vendored rltorch builder (initialisation: xavier-uniform
weights, zero biases — that builder is not in /root/reference, so its exact initialiser is not claimed).  The known crashes of
the reference (the 14-input bc_actor, train_episode's missing env) are not reproduced.
"""
import os

import numpy as np
import torch

from ...utils.buffer import DeviceReplay, device
from .. import sac_engine as SE


class _Writer:
    """SummaryWriter stand-in when tensorboard is absent (agent.writer is used by train_sac.py:217)."""

    def add_scalar(self, *a, **k):
        pass


class DeviceMemory(DeviceReplay):
    """rltorch MultiStepMemory's append / sample / len on the device ring (multi_step = 1, SAC/agent.py:121-124)."""

    def __init__(self, capacity):
        super().__init__(int(capacity), device)
        self._len, self._pos = 0, 0

    def append(self, state, action, reward, next_state, done, episode_done=None):  # train_sac.py:242
        row = np.zeros(32, np.float32)
        row[0:13], row[13:17], row[17:30], row[30], row[31] = state, action, next_state, reward, float(done)
        self.ring[self._pos] = torch.from_numpy(row).to(self.device)
        self._pos = (self._pos + 1) % self.capacity
        self._len = min(self._len + 1, self.capacity)
        self.total += 1

    def __len__(self):
        return self._len

    def sample_indices(self, n):
        return np.random.randint(low=0, high=self._len, size=n)  # rltorch memories draw with np.random.randint

    def sample(self, n):
        rows = self.ring[torch.as_tensor(self.sample_indices(n), device=self.device)]
        return rows[:, 0:13], rows[:, 13:17], rows[:, 30:31], rows[:, 17:30], rows[:, 31:32]


def _xavier_mlp(n_in, n_out):
    sd = {}
    for key, (o, i) in (("0", (256, n_in)), ("2", (512, 256)), ("4", (n_out, 512))):
        w = torch.empty(o, i)
        torch.nn.init.xavier_uniform_(w)
        sd[key + ".weight"], sd[key + ".bias"] = w, torch.zeros(o)
    return sd


class SacAgent:
    def __init__(self, observation_space, action_space, log_dir, num_steps=3000000, batch_size=256, lr=0.0003, hidden_units=[256, 256],
                 memory_size=1e6, gamma=0.99, tau=0.005, imitative=False, entropy_tuning=True, ent_coef=0.2, multi_step=1, per=False,
                 alpha=0.6, beta=0.4, beta_annealing=0.0001, grad_clip=None, updates_per_step=1, start_steps=10000, log_interval=300,
                 target_update_interval=3, eval_interval=1000, cuda=True):
        if tuple(observation_space.shape) != (13,) or tuple(action_space.shape) != (4,) or list(hidden_units) != [256, 512]:
            raise NotImplementedError("the HIP kernels are built for train_sac.py's shape: 13 / 4 / hidden [256, 512]")
        if imitative or per or multi_step != 1 or not entropy_tuning or grad_clip is not None:
            raise NotImplementedError("only the configuration train_sac.py runs (non-imitative, uniform replay, entropy tuning) is built")
        self.observation_space, self.action_space = observation_space, action_space
        self.device = device
        self.eng = SE.SacEngine(batch=batch_size, lr=lr, gamma=gamma ** multi_step, tau=tau, target_entropy=-float(np.prod(action_space.shape)),
                                target_update_interval=target_update_interval, device=device)
        self.eng.load_params(_xavier_mlp(13, 8), _xavier_mlp(17, 1), _xavier_mlp(17, 1))
        self.memory = DeviceMemory(memory_size)
        self.expert_memory = None
        self.log_dir = log_dir
        self.model_dir, self.summary_dir, self.plot_dir = (os.path.join(log_dir, d) for d in ("model", "summary", "plot"))
        for d in (self.model_dir, self.summary_dir, self.plot_dir):
            os.makedirs(d, exist_ok=True)
        try:
            from torch.utils.tensorboard import SummaryWriter

            self.writer = SummaryWriter(log_dir=self.summary_dir)
        except Exception:
            self.writer = _Writer()
        self.steps = self.episodes = 0
        self.batch_size, self.start_steps, self.tau = batch_size, start_steps, tau
        self.gamma_n, self.entropy_tuning, self.log_interval = gamma ** multi_step, entropy_tuning, log_interval
        self.target_update_interval = target_update_interval

    @property
    def learning_steps(self):
        return self.eng.learning_steps

    @property
    def alpha(self):
        return self.eng.alpha_state[3]

    def is_update(self):  # agent.py:170-172
        return len(self.memory) > self.batch_size and self.steps >= self.start_steps

    def explore(self, state):  # agent.py:183-188
        obs = torch.as_tensor(np.asarray(state, np.float32).reshape(1, 13)).to(device)
        eps = torch.randn(1, 4).to(device)  # Normal.rsample draws from torch's generator
        return self.eng.act(obs, eps=eps)[0].cpu().numpy().reshape(-1)

    def exploit(self, state):  # agent.py:191-196
        obs = torch.as_tensor(np.asarray(state, np.float32).reshape(1, 13)).to(device)
        return self.eng.act(obs, explore=False)[0].cpu().numpy().reshape(-1)

    def act(self, state):  # agent.py:175-180
        return self.action_space.sample() if self.start_steps > self.steps else self.explore(state)

    def learn(self, if_expert, expert_num=None, expert_data=None):  # agent.py:276-359
        B = self.batch_size
        if if_expert and expert_num:
            # expert rows were drawn by the caller (expert_memory.sample(expert_num), train_sac.py:272-273) and come last
            main = self.memory.ring[torch.as_tensor(self.memory.sample_indices(B - expert_num), device=device)]
            s, a, r, ns, d = expert_data
            rows = torch.cat([main, torch.cat([s, a, ns, r, d], 1).to(device)], 0).contiguous()
        else:
            rows = self.memory.ring[torch.as_tensor(self.memory.sample_indices(B), device=device)].contiguous()
        self.eng.rows.copy_(rows.reshape(-1))
        self.eng.learn(torch.randn(B, 4).to(device), torch.randn(B, 4).to(device))
        if self.eng.learning_steps % self.log_interval == 0:
            q1, q2, pl, el, ent, alpha = self.eng.losses_host()
            for k, v in (("loss/Q1", q1), ("loss/Q2", q2), ("loss/policy", pl), ("stats/alpha", alpha), ("stats/entropy", ent)):
                self.writer.add_scalar(k, v, self.eng.learning_steps)

    def save_models(self, ajan):  # agent.py:440-444
        self.eng.save_models(self.model_dir, ajan)
