"""oracle/hirl_oracle.py — CPU ORACLE for the HIRL (TD3+BC) update.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
(hirl4ucav_amd/) never does.

A plain fp32 PyTorch-on-CPU restatement of the reference's update path, written from the reference's algorithm
(not from its classes): functional networks over explicit parameter dicts, autograd for the gradients, hand-written
Adam and Polyak.  PINNED: tests/test_oracle_hirl.py checks it against golden vectors recorded from the reference's own
``hirl.agents.HIRL.Agent.learn`` / ``chooseAction*`` (tests/golden/gen_hirl_golden.py -> tests/golden/hirl_learn_*.npz).

    actor_forward        <- Actor.forward                 hirl/agents/HIRL.py:126-140   (U1)
    critic_forward / q1  <- Critic.forward / onlyQ1       hirl/agents/HIRL.py:55-97     (U2)
    polyak               <- soft_update                   hirl/agents/HIRL.py:11-13     (U3)
    HirlOracle.learn     <- Agent.learn                   hirl/agents/HIRL.py:221-334   (U7-U11)
    HirlOracle.choose_action <- chooseAction*             hirl/agents/HIRL.py:192-212   (U5)
    adam_step            <- torch.optim.Adam defaults (lr, betas (0.9, 0.999), eps 1e-8)  HIRL.py:50,123
    TD3 variant (leaky_relu 0.01, no BC)  <- hirl/agents/TD3.py:201-260 (U12) via slope=0.01, bc off
    HirlOracle.bc_train_actor <- BC.Agent.train_actor  hirl/agents/BC.py:160-185
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

ACTOR_KEYS = ("full1.weight", "full1.bias", "layernorm1.weight", "layernorm1.bias", "full2.weight", "full2.bias",
              "layernorm2.weight", "layernorm2.bias", "final.weight", "final.bias")
CRITIC_KEYS = ("full1.weight", "full1.bias", "layernorm1.weight", "layernorm1.bias", "full2.weight", "full2.bias",
               "layernorm2.weight", "layernorm2.bias", "final1.weight", "final1.bias",
               "full3.weight", "full3.bias", "layernorm3.weight", "layernorm3.bias", "full4.weight", "full4.bias",
               "layernorm4.weight", "layernorm4.bias", "final2.weight", "final2.bias")


def _uniform(rng, bound, shape):
    return rng.uniform(-bound, bound, shape).astype(np.float32)


def init_actor(rng, sd=13, ad=4, h1=256, h2=512, perturb_ln=True):
    """Initialisation with the reference's bounds: hidden weights kaiming_uniform_(a=0.01, fan_in, 'relu') =
    U(+-sqrt(6/fan_in)) (HIRL.py:111,116); biases and `final` = nn.Linear default U(+-1/sqrt(fan_in)).  LayerNorm
    affine parameters start at (1, 0) in the reference; perturb_ln draws them near (1, 0) instead so that parity
    tests exercise them."""
    p = {}
    p["full1.weight"] = _uniform(rng, math.sqrt(6.0 / sd), (h1, sd))
    p["full1.bias"] = _uniform(rng, 1 / math.sqrt(sd), (h1,))
    p["full2.weight"] = _uniform(rng, math.sqrt(6.0 / h1), (h2, h1))
    p["full2.bias"] = _uniform(rng, 1 / math.sqrt(h1), (h2,))
    p["final.weight"] = _uniform(rng, 1 / math.sqrt(h2), (ad, h2))
    p["final.bias"] = _uniform(rng, 1 / math.sqrt(h2), (ad,))
    for k, h in (("layernorm1", h1), ("layernorm2", h2)):
        p[k + ".weight"] = (1 + (rng.normal(0, 0.05, h) if perturb_ln else 0)).astype(np.float32) * np.ones(h, np.float32)
        p[k + ".bias"] = (rng.normal(0, 0.05, h) if perturb_ln else np.zeros(h)).astype(np.float32)
    return {k: p[k] for k in ACTOR_KEYS}


def init_critic(rng, sd=13, ad=4, h1=256, h2=512, perturb_ln=True):
    p = {}
    for a, b, f, ln1, ln2 in (("full1", "full2", "final1", "layernorm1", "layernorm2"), ("full3", "full4", "final2", "layernorm3", "layernorm4")):
        p[a + ".weight"] = _uniform(rng, math.sqrt(6.0 / (sd + ad)), (h1, sd + ad))
        p[a + ".bias"] = _uniform(rng, 1 / math.sqrt(sd + ad), (h1,))
        p[b + ".weight"] = _uniform(rng, math.sqrt(6.0 / h1), (h2, h1))
        p[b + ".bias"] = _uniform(rng, 1 / math.sqrt(h1), (h2,))
        p[f + ".weight"] = _uniform(rng, 1 / math.sqrt(h2), (1, h2))
        p[f + ".bias"] = _uniform(rng, 1 / math.sqrt(h2), (1,))
        for k, h in ((ln1, h1), (ln2, h2)):
            p[k + ".weight"] = (1 + (rng.normal(0, 0.05, h) if perturb_ln else 0)).astype(np.float32) * np.ones(h, np.float32)
            p[k + ".bias"] = (rng.normal(0, 0.05, h) if perturb_ln else np.zeros(h)).astype(np.float32)
    return {k: p[k] for k in CRITIC_KEYS}


def to_torch(p, grad=False, device="cpu"):
    return {k: torch.tensor(np.asarray(v), dtype=torch.float32, device=device, requires_grad=grad) for k, v in p.items()}


def _act(x, slope):
    return F.relu(x) if slope == 0.0 else F.leaky_relu(x, slope)


def _ln(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


class NoLayerNorm:
    """context manager: the networks' `layerNorm=False` branches (HIRL.py:70-80,92-97,135-138): both LayerNorms are skipped (the modules
    stay in the state_dict, untrained)"""

    def __enter__(self):
        global _ln
        self._orig = _ln
        _ln = lambda x, w, b: x  # noqa: E731
        return self

    def __exit__(self, *exc):
        global _ln
        _ln = self._orig


class _Maybe:
    def __init__(self, cm):
        self.cm = cm

    def __enter__(self):
        return self.cm.__enter__() if self.cm is not None else None

    def __exit__(self, *exc):
        return self.cm.__exit__(*exc) if self.cm is not None else False


def _linear2(h, w, b):
    """the 256 -> 512 layer.  fp32 F.linear — or, inside `with Bf16Layer2():`, the ROUNDED-OPERAND form the bf16 update path is tested
    against (tests/test_bf16_update_gpu.py)."""
    return F.linear(h, w, b)


class _RoundedLinear2(torch.autograd.Function):
    """z = bf16(h) bf16(W)^T + b;  dh = bf16(dz) bf16(W);  dW = bf16(dz)^T bf16(h);  db = sum(dz) — products exact, sums in fp64
    (the reference value for fp32 accumulation in any order); everything outside these three products stays fp32."""

    @staticmethod
    def forward(ctx, h, w, b):
        hb, wb = h.to(torch.bfloat16).double(), w.to(torch.bfloat16).double()
        ctx.save_for_backward(hb, wb)
        return (hb @ wb.t()).float() + b

    @staticmethod
    def backward(ctx, g):
        hb, wb = ctx.saved_tensors
        gb = g.to(torch.bfloat16).double()
        return (gb @ wb).float(), (gb.t() @ hb).float(), g.sum(0)


class Bf16Layer2:
    """context manager: every _linear2 inside runs as _RoundedLinear2 (BASELINE.json configs[4] "bf16 actor/critic")"""

    def __enter__(self):
        global _linear2
        self._orig = _linear2
        _linear2 = _RoundedLinear2.apply
        return self

    def __exit__(self, *exc):
        global _linear2
        _linear2 = self._orig


def actor_forward(p, x, slope=0.0):
    """tanh(W3 act(LN2(W2 act(LN1(W1 x + b1)) + b2)) + b3)   HIRL.py:126-140"""
    h = _act(_ln(F.linear(x, p["full1.weight"], p["full1.bias"]), p["layernorm1.weight"], p["layernorm1.bias"]), slope)
    h = _act(_ln(_linear2(h, p["full2.weight"], p["full2.bias"]), p["layernorm2.weight"], p["layernorm2.bias"]), slope)
    return torch.tanh(F.linear(h, p["final.weight"], p["final.bias"]))


def _q_head(p, sa, names, slope):
    a, ln1, b, ln2, f = names
    h = _act(_ln(F.linear(sa, p[a + ".weight"], p[a + ".bias"]), p[ln1 + ".weight"], p[ln1 + ".bias"]), slope)
    h = _act(_ln(_linear2(h, p[b + ".weight"], p[b + ".bias"]), p[ln2 + ".weight"], p[ln2 + ".bias"]), slope)
    return F.linear(h, p[f + ".weight"], p[f + ".bias"])


_Q1 = ("full1", "layernorm1", "full2", "layernorm2", "final1")
_Q2 = ("full3", "layernorm3", "full4", "layernorm4", "final2")


def critic_q1(p, s, a, slope=0.0):  # HIRL.py:82-97
    return _q_head(p, torch.cat([s, a], 1), _Q1, slope)


def critic_forward(p, s, a, slope=0.0):  # HIRL.py:55-80
    sa = torch.cat([s, a], 1)
    return _q_head(p, sa, _Q1, slope), _q_head(p, sa, _Q2, slope)


class Adam:
    """torch.optim.Adam with default arguments, restated (single-tensor form)."""

    def __init__(self, params, lr, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps = lr, b1, b2, eps
        self.t = 0
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}

    def step(self, params, grads):
        self.t += 1
        bc1 = 1.0 - self.b1 ** self.t
        bc2 = 1.0 - self.b2 ** self.t
        with torch.no_grad():
            for k, p in params.items():
                g = grads[k]
                self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
                self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
                p.addcdiv_(self.m[k], denom, value=-(self.lr / bc1))


def _grads(loss, params, keys, retain_graph=False):
    """d loss / d params[k]; a parameter the loss does not reach (the LayerNorm modules of a layerNorm=False network: torch.optim.Adam
    skips their None gradients) gets a zero gradient, which leaves it and its moments unchanged in Adam.step above"""
    g = torch.autograd.grad(loss, [params[k] for k in keys], retain_graph=retain_graph, allow_unused=True)
    return [torch.zeros_like(params[k]) if x is None else x for k, x in zip(keys, g)]


def polyak(target, source, tau):  # HIRL.py:11-13
    with torch.no_grad():
        for k in target:
            target[k].copy_(target[k] * (1.0 - tau) + source[k] * tau)


class HirlOracle:
    """State and update rule of hirl.agents.HIRL.Agent (and TD3.Agent with slope=0.01, use_bc=False)."""

    def __init__(self, actor, critic, bc_actor=None, lr_actor=1e-3, lr_critic=1e-3, tau=0.005, gamma=0.99, slope=0.0,
                 use_bc=True, device="cpu", layer_norm=True):
        self.layer_norm = bool(layer_norm)  # False: Agent(..., layerNorm=False, ...)
        # device: "cpu" for every parity test; bench.py's baseline leg also runs the same eager ops on the GPU (b2_eager_rocm_learn)
        self.dev = torch.device(device)
        self.actor = to_torch(actor, True, self.dev)
        self.critic = to_torch(critic, True, self.dev)
        self.target_actor = to_torch(actor, False, self.dev)   # hard_update HIRL.py:172
        self.target_critic = to_torch(critic, False, self.dev)  # HIRL.py:176
        self.bc_actor = to_torch(bc_actor, False, self.dev) if bc_actor is not None else None
        self.opt_actor, self.opt_critic = Adam(self.actor, lr_actor), Adam(self.critic, lr_critic)
        self.tau, self.gamma, self.slope, self.use_bc = tau, gamma, slope, use_bc
        self.noise_clamp, self.loss_lambda, self.target_update_freq = 0.5, 10000.0, 3  # HIRL.py:162,182-183
        self.actor_trainable, self.update_count = True, 0
        self.bc_weight = 0.0
        self.actor_loss = self.bc_loss = self.rl_loss = 0.0
        self.bc_fire_loss = 0.0
        self.last_grads = {}
        # True: grad(actor_loss) formed as w grad(bc_loss) + (1 - w) grad(rl_loss) from two backward passes (what a sharded run exchanges,
        # SURVEY.md 8e; equal in exact arithmetic).  The rounded-operand bf16 check needs it: bf16(w dz) != w bf16(dz).
        self.split_actor_grads = False

    def choose_action(self, state, noise=None):
        """clamp(actor(s) + noise, -1, 1); noise None = chooseActionNoNoise   HIRL.py:192-212"""
        with torch.no_grad(), _Maybe(None if self.layer_norm else NoLayerNorm()):
            a = actor_forward(self.actor, torch.as_tensor(state, dtype=torch.float32, device=self.dev), self.slope)
            if noise is not None:
                a = (a + torch.as_tensor(noise, dtype=torch.float32, device=self.dev)).clamp(-1, 1)
        return a.cpu().numpy()

    def learn(self, batch, bc_batch, noise, bc_weight_now=0.0, bc_warm_up_weight=0.0):
        with _Maybe(None if self.layer_norm else NoLayerNorm()):
            return self._learn(batch, bc_batch, noise, bc_weight_now, bc_warm_up_weight)

    def _learn(self, batch, bc_batch, noise, bc_weight_now=0.0, bc_warm_up_weight=0.0):
        """batch = (s[B,13], a[B,4], s'[B,13], r[B], d[B]) already mixed buffer ++ expert rows (HIRL.py:223-243);
        bc_batch = (s_bc[B,13], a_bc[B,4]) (HIRL.py:248-251); noise = the ONE (4,) N(0, 0.2^2) draw shared by the
        whole batch (HIRL.py:265), unclamped.  Returns the reference's 6-tuple (HIRL.py:334)."""
        s, a, ns, r, d = (torch.as_tensor(x, dtype=torch.float32, device=self.dev) for x in batch)
        # U8: TD target  HIRL.py:259-274
        with torch.no_grad():
            na = actor_forward(self.target_actor, ns, self.slope)
            eps = torch.as_tensor(noise, dtype=torch.float32, device=self.dev).clamp(-self.noise_clamp, self.noise_clamp)
            na = (na + eps).clamp(-1, 1)
            tq1, tq2 = critic_forward(self.target_critic, ns, na, self.slope)
            y = r.reshape(-1, 1) + self.gamma * torch.min(tq1, tq2) * (1 - d).reshape(-1, 1)
        # U9: critic step  HIRL.py:276-288
        q1, q2 = critic_forward(self.critic, s, a, self.slope)
        critic_loss = F.mse_loss(q1, y) + F.mse_loss(q2, y)
        keys = list(self.critic)
        grads = dict(zip(keys, _grads(critic_loss, self.critic, keys)))
        self.last_grads["critic"] = {k: g.clone() for k, g in grads.items()}
        self.opt_critic.step(self.critic, grads)
        # U10: delayed actor step with the UPDATED critic  HIRL.py:291-330
        if self.actor_trainable:
            self.bc_weight = bc_weight_now
            pi = actor_forward(self.actor, s, self.slope)
            rl_q = critic_q1(self.critic, s, pi, self.slope)
            rl_loss = -rl_q.mean()
            if self.use_bc:
                if self.bc_weight == 100:  # "soft" sentinel  HIRL.py:299-306
                    with torch.no_grad():
                        soft_q = critic_q1(self.critic, s, actor_forward(self.bc_actor, s, self.slope), self.slope)
                        self.bc_weight = (soft_q > rl_q).float().mean().item()
                    self.bc_weight += bc_warm_up_weight
                if self.bc_weight > 1:
                    self.bc_weight = 1
                bs, ba = (torch.as_tensor(x, dtype=torch.float32, device=self.dev) for x in bc_batch)
                bc_pred = actor_forward(self.actor, bs, self.slope)
                bc_loss = F.mse_loss(bc_pred, ba) * self.loss_lambda                     # HIRL.py:310-311
                self.bc_fire_loss = F.mse_loss(bc_pred[:, 3].detach(), ba[:, 3]).item() * self.loss_lambda  # :317-319
                actor_loss = bc_loss * self.bc_weight + rl_loss * (1 - self.bc_weight)  # :321
                self.bc_loss = bc_loss.item()
            else:  # TD3.py:233-236
                actor_loss = rl_loss
            akeys = list(self.actor)
            if self.split_actor_grads and self.use_bc:
                g_bc = _grads(bc_loss, self.actor, akeys, retain_graph=True)
                g_rl = _grads(rl_loss, self.actor, akeys)
                w = float(self.bc_weight)
                agrads = {k: w * a + (1.0 - w) * b for k, a, b in zip(akeys, g_bc, g_rl)}
            else:
                agrads = dict(zip(akeys, _grads(actor_loss, self.actor, akeys)))
            self.last_grads["actor"] = {k: g.clone() for k, g in agrads.items()}
            self.opt_actor.step(self.actor, agrads)
            self.actor_loss, self.rl_loss = actor_loss.item(), rl_loss.item()
            self.update_count += 1
            if self.update_count % self.target_update_freq == 0:  # HIRL.py:327-330
                polyak(self.target_critic, self.critic, self.tau)
                polyak(self.target_actor, self.actor, self.tau)
        self.actor_trainable = not self.actor_trainable  # HIRL.py:332
        return critic_loss.item(), self.actor_loss, self.bc_loss, self.rl_loss, self.bc_fire_loss, self.bc_weight


def bc_train_actor(o, bc_batch):
    """BC.Agent.train_actor (BC.py:160-185) on an oracle built with slope=0.01: mse(actor(s), a), backward, Adam."""
    bs, ba = (torch.as_tensor(x, dtype=torch.float32) for x in bc_batch)
    with _Maybe(None if o.layer_norm else NoLayerNorm()):
        loss = F.mse_loss(actor_forward(o.actor, bs, o.slope), ba)
    keys = list(o.actor)
    grads = dict(zip(keys, _grads(loss, o.actor, keys)))
    o.last_grads["actor"] = {k: g.clone() for k, g in grads.items()}
    o.opt_actor.step(o.actor, grads)
    return loss.item()


# ---- flat parameter layout shared by the tests (order = the reference's state_dict order) ----------------------
def flatten(p, keys):
    return np.concatenate([np.asarray(p[k].detach().numpy() if torch.is_tensor(p[k]) else p[k], np.float32).ravel() for k in keys])
