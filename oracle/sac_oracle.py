"""oracle/sac_oracle.py — CPU ORACLE for the SAC update.  TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench's cpu leg).

Plain fp32 torch-on-CPU restatement of hirl/agents/SAC (non-imitative branch, the one train_sac.py uses):

    policy_forward / sample  <- GaussianPolicy.forward / sample       SAC/model.py:63-82       (U14)
    q_forward                <- TwinnedQNetwork.forward               SAC/model.py:41-45
    SacOracle.learn          <- SacAgent.learn                        SAC/agent.py:276-327     (U13)
       target                <- calc_target_q                         SAC/agent.py:202-210
       critic losses         <- calc_critic_loss                      SAC/agent.py:361-374
       policy loss           <- calc_policy_loss (non-imitative)      SAC/agent.py:376-406
       entropy loss          <- calc_entropy_loss                     SAC/agent.py:408-414
    explore / exploit        <- SacAgent.explore / exploit            SAC/agent.py:183-196

PINNED for the loss math: tests/test_oracle_sac.py replays golden vectors recorded from the reference's own SacAgent.learn
(tests/golden/gen_sac_golden.py).  **Parity UNPINNED for two things that live in the un-vendored `rltorch` dependency**
(ku2482/rltorch, no version pinned in setup.py:7-14): the network builder `create_linear_network(..., initializer='xavier')`
(SAC/model.py:4,21,58) and the replay memories (SAC/agent.py:6,115-124).  The golden run substitutes the published builder
shape — Linear(in,256) ReLU Linear(256,512) ReLU Linear(512,out), which reproduces the parameter counts SURVEY.md 2.1 lists
(139,272 / 273,410) — and injected weights / minibatches / noise, so initialisation and sampling order are not claimed.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .hirl_oracle import Adam

MLP_KEYS = ("0.weight", "0.bias", "2.weight", "2.bias", "4.weight", "4.bias")  # nn.Sequential(Linear, ReLU, Linear, ReLU, Linear)
LOG_STD_MAX, LOG_STD_MIN, EPS = 2.0, -20.0, 1e-6  # SAC/model.py:49-51


def init_mlp(rng, n_in, n_out, h1=256, h2=512):
    """xavier-uniform weights, zero biases (what rltorch's 'xavier' initializer is understood to do — unpinned)."""
    def xav(o, i):
        b = math.sqrt(6.0 / (i + o))
        return rng.uniform(-b, b, (o, i)).astype(np.float32)
    return {"0.weight": xav(h1, n_in), "0.bias": rng.normal(0, 0.05, h1).astype(np.float32),
            "2.weight": xav(h2, h1), "2.bias": rng.normal(0, 0.05, h2).astype(np.float32),
            "4.weight": xav(n_out, h2), "4.bias": rng.normal(0, 0.05, n_out).astype(np.float32)}


def _act(x, slope=0.0):
    """the builder's ReLU; a module-level hook so that tests can record pre-activations / pick the subgradient at a kink"""
    return F.relu(x)


def mlp(p, x):
    h = _act(F.linear(x, p["0.weight"], p["0.bias"]))
    h = _act(F.linear(h, p["2.weight"], p["2.bias"]))
    return F.linear(h, p["4.weight"], p["4.bias"])


def policy_forward(p, s):  # model.py:63-68
    mean, log_std = torch.chunk(mlp(p, s), 2, dim=-1)
    return mean, torch.clamp(log_std, min=LOG_STD_MIN, max=LOG_STD_MAX)


def sample(p, s, eps):
    """actions, entropies, tanh(means) with the standard-normal draw `eps` injected   model.py:69-82"""
    mean, log_std = policy_forward(p, s)
    std = log_std.exp()
    x = mean + std * eps  # Normal.rsample
    a = torch.tanh(x)
    log_prob = (-((x - mean) ** 2) / (2 * std ** 2) - log_std - math.log(math.sqrt(2 * math.pi))) - torch.log(1 - a.pow(2) + EPS)
    return a, -log_prob.sum(dim=1, keepdim=True), torch.tanh(mean)


def to_t(p, grad=False):
    return {k: torch.tensor(np.asarray(v), dtype=torch.float32, requires_grad=grad) for k, v in p.items()}


class SacOracle:
    def __init__(self, policy, q1, q2, lr=1e-3, gamma=0.99, tau=0.005, target_entropy=-4.0, target_update_interval=3):
        self.policy, self.q1, self.q2 = to_t(policy, True), to_t(q1, True), to_t(q2, True)
        self.q1_t, self.q2_t = to_t(q1), to_t(q2)  # hard_update, agent.py:92
        self.log_alpha = torch.zeros(1, requires_grad=True)  # agent.py:106
        self.alpha = self.log_alpha.exp().detach()
        self.opt_pi, self.opt_q1, self.opt_q2 = Adam(self.policy, lr), Adam(self.q1, lr), Adam(self.q2, lr)
        self.opt_alpha = Adam({"a": self.log_alpha}, lr)
        self.gamma, self.tau, self.target_entropy, self.interval = gamma, tau, target_entropy, target_update_interval
        self.learning_steps = 0
        self.last_grads = {}

    def explore(self, s, eps):  # agent.py:183-188
        with torch.no_grad():
            return sample(self.policy, torch.as_tensor(s, dtype=torch.float32).reshape(-1, 13), torch.as_tensor(eps, dtype=torch.float32).reshape(-1, 4))[0].numpy()

    def exploit(self, s):  # agent.py:191-196
        with torch.no_grad():
            return torch.tanh(policy_forward(self.policy, torch.as_tensor(s, dtype=torch.float32).reshape(-1, 13))[0]).numpy()

    def learn(self, batch, eps_next, eps_cur):
        """batch = (s, a, r[B], s', d[B]); eps_next / eps_cur = the [B, 4] standard-normal draws of the two policy.sample calls.
        Returns (q1_loss, q2_loss, policy_loss, entropy_loss, mean entropy, alpha after the step)."""
        s, a, r, ns, d = (torch.as_tensor(x, dtype=torch.float32) for x in batch)
        r, d = r.reshape(-1, 1), d.reshape(-1, 1)
        e1, e2 = torch.as_tensor(eps_next, dtype=torch.float32), torch.as_tensor(eps_cur, dtype=torch.float32)
        self.learning_steps += 1
        if self.learning_steps % self.interval == 0:  # agent.py:278-279 — BEFORE the update
            with torch.no_grad():
                for t, src in ((self.q1_t, self.q1), (self.q2_t, self.q2)):
                    for k in t:
                        t[k].copy_(t[k] * (1.0 - self.tau) + src[k] * self.tau)
        sa = torch.cat([s, a], 1)
        with torch.no_grad():  # calc_target_q
            na, nh, _ = sample(self.policy, ns, e1)
            nsa = torch.cat([ns, na], 1)
            next_q = torch.min(mlp(self.q1_t, nsa), mlp(self.q2_t, nsa)) + self.alpha * nh
            y = r + (1.0 - d) * self.gamma * next_q
        q1_loss = torch.mean((mlp(self.q1, sa) - y).pow(2))
        q2_loss = torch.mean((mlp(self.q2, sa) - y).pow(2))
        for name, net, opt, loss in (("q1", self.q1, self.opt_q1, q1_loss), ("q2", self.q2, self.opt_q2, q2_loss)):
            keys = list(net)
            g = dict(zip(keys, torch.autograd.grad(loss, [net[k] for k in keys])))
            self.last_grads[name] = {k: v.clone() for k, v in g.items()}
            opt.step(net, g)
        # calc_policy_loss with the UPDATED critics
        pa, ent, _ = sample(self.policy, s, e2)
        psa = torch.cat([s, pa], 1)
        q = torch.min(mlp(self.q1, psa), mlp(self.q2, psa))
        policy_loss = torch.mean(-q - self.alpha * ent)
        keys = list(self.policy)
        g = dict(zip(keys, torch.autograd.grad(policy_loss, [self.policy[k] for k in keys])))
        self.last_grads["policy"] = {k: v.clone() for k, v in g.items()}
        self.opt_pi.step(self.policy, g)
        # calc_entropy_loss
        entropy_loss = -torch.mean(self.log_alpha * (self.target_entropy - ent).detach())
        ga = torch.autograd.grad(entropy_loss, [self.log_alpha])[0]
        self.opt_alpha.step({"a": self.log_alpha}, {"a": ga})
        self.alpha = self.log_alpha.exp().detach()
        return q1_loss.item(), q2_loss.item(), policy_loss.item(), entropy_loss.item(), ent.mean().item(), self.alpha.item()


def flatten(p):
    return np.concatenate([(p[k].detach().numpy() if torch.is_tensor(p[k]) else np.asarray(p[k])).ravel() for k in MLP_KEYS]).astype(np.float32)
