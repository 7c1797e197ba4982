#!/usr/bin/env python3
"""Generate golden vectors for the SAC update by running the REFERENCE SacAgent.learn (hirl/agents/SAC/agent.py).

Development container only.  `rltorch` (un-vendored, unpinned), `tensorboard` and `gym` are absent from this image: rltorch's
network builder is replaced by the published shape Sequential(Linear, ReLU, Linear, ReLU, Linear) and its memory by a stub that
returns the minibatch the generator chose, so what these vectors pin is the reference's LOSS MATH and update order
(Polyak-before-update every 3rd call, two critic optimisers, policy step with the updated critics, log-alpha step), not rltorch's
initialiser or sampling order.

    python tests/golden/gen_sac_golden.py   ->  tests/golden/sac_learn.npz
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REF, "hirl"))

# ---- stubs ---------------------------------------------------------------------------------------------------------
rl = types.ModuleType("rltorch")
rl.network = types.ModuleType("rltorch.network")
rl.memory = types.ModuleType("rltorch.memory")


def create_linear_network(input_dim, output_dim, hidden_units=[], hidden_activation="relu", output_activation=None, initializer="xavier"):
    layers, units = [], input_dim
    for nxt in hidden_units:
        layers += [nn.Linear(units, nxt), nn.ReLU()]
        units = nxt
    layers.append(nn.Linear(units, output_dim))
    return nn.Sequential(*layers)


class _Memory:
    def __init__(self, *a, **k):
        self.next_batch = None

    def sample(self, n):
        return self.next_batch

    def __len__(self):
        return 10 ** 6


rl.network.create_linear_network = create_linear_network
rl.memory.MultiStepMemory = _Memory
rl.memory.PrioritizedMemory = _Memory
sys.modules.update({"rltorch": rl, "rltorch.network": rl.network, "rltorch.memory": rl.memory})
tb = types.ModuleType("torch.utils.tensorboard")


class SummaryWriter:
    def __init__(self, *a, **k):
        pass

    def add_scalar(self, *a, **k):
        pass


tb.SummaryWriter = SummaryWriter
sys.modules["torch.utils.tensorboard"] = tb

from oracle import sac_oracle as S  # noqa: E402
from tests import _hirl_data as D  # noqa: E402

import agents.SAC.agent as ref_sac  # noqa: E402
import torch.distributions.normal as tdn  # noqa: E402

torch.set_num_threads(1)


def main():
    rng = np.random.default_rng(31)
    params = {"policy": S.init_mlp(rng, 13, 8), "q1": S.init_mlp(rng, 17, 1), "q2": S.init_mlp(rng, 17, 1)}
    data = D.make_data(D.DATA_SEED)
    box = lambda n: types.SimpleNamespace(shape=(n,))  # noqa: E731
    agent = ref_sac.SacAgent(observation_space=box(13), action_space=box(4), log_dir=tempfile.mkdtemp(), batch_size=128, lr=1e-3,
                             hidden_units=[256, 512], memory_size=2e5, gamma=0.99, tau=0.005, cuda=False)  # train_sac.py:214-215
    sd = lambda p: {k: torch.tensor(v) for k, v in p.items()}  # noqa: E731
    agent.policy.policy.load_state_dict(sd(params["policy"]))
    for net in (agent.critic, agent.critic_target):
        net.Q1.Q.load_state_dict(sd(params["q1"]))
        net.Q2.Q.load_state_dict(sd(params["q2"]))
    draws = []

    def fake_standard_normal(shape, dtype, device):
        e = torch.tensor(rng.normal(0, 1, tuple(shape)).astype(np.float32))
        draws.append(e.numpy().copy())
        return e

    tdn._standard_normal = fake_standard_normal
    K = 8
    idx, outs, prb = [], [], []
    for k in range(K):
        i = rng.choice(D.N_REPLAY, 128, replace=False)
        rows = data["replay"][i]
        agent.memory.next_batch = (torch.tensor(rows[:, 0:13]), torch.tensor(rows[:, 13:17]), torch.tensor(rows[:, 30:31]),
                                   torch.tensor(rows[:, 17:30]), torch.tensor(rows[:, 31:32]))  # (s, a, r, s', done)  train_sac.py:242
        # capture the losses the reference computes inside learn()
        rec = {}
        orig_c, orig_p, orig_e = agent.calc_critic_loss, agent.calc_policy_loss, agent.calc_entropy_loss

        def cc(b, w, orig=orig_c):
            out = orig(b, w)
            rec["q1"], rec["q2"] = out[0].item(), out[1].item()
            return out

        def cp(b, w, orig=orig_p):
            out = orig(b, w)
            rec["pi"], rec["ent"] = out[0].item(), out[1].detach().mean().item()
            return out

        def ce(e, w, orig=orig_e):
            out = orig(e, w)
            rec["el"] = out.item()
            return out

        agent.calc_critic_loss, agent.calc_policy_loss, agent.calc_entropy_loss = cc, cp, ce
        agent.learn(False)
        agent.calc_critic_loss, agent.calc_policy_loss, agent.calc_entropy_loss = orig_c, orig_p, orig_e
        idx.append(i.astype(np.int32))
        outs.append([rec["q1"], rec["q2"], rec["pi"], rec["el"], rec["ent"], agent.alpha.item()])
        row = []
        for net in (agent.policy.policy, agent.critic.Q1.Q, agent.critic.Q2.Q, agent.critic_target.Q1.Q, agent.critic_target.Q2.Q):
            flat = np.concatenate([v.detach().numpy().ravel() for v in net.state_dict().values()]).astype(np.float64)
            row.append((np.abs(flat).sum(), flat[D.probe_index(flat.size)]))
        prb.append(row)
    np.savez_compressed(os.path.join(HERE, "sac_learn.npz"), idx=np.asarray(idx), eps=np.asarray(draws, np.float32).reshape(K, 2, 128, 4),
                        out=np.asarray(outs, np.float64), probe_abs=np.asarray([[p[0] for p in r] for r in prb]),
                        probe_val=np.asarray([[p[1] for p in r] for r in prb], np.float32), data_checksum=D.checksum(data),
                        param_checksum=D.checksum(params))
    print("sac q1", [round(o[0], 3) for o in outs[:4]], "pi", [round(o[2], 4) for o in outs[:4]], "alpha", [round(o[5], 5) for o in outs])


if __name__ == "__main__":
    main()
