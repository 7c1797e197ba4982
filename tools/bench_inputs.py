"""Synthetic inputs of bench.py: the expert set and the seeded-init networks, shared by the GPU loop and the CPU baseline legs."""
import math

import numpy as np


def synthetic_expert(rng, n=20000):
    """states U(-1,1)^13 with cols 7,8 in {+-1}, col 12 in [0, 0.2]; actions U(-1,1)^3 ++ fire +-1, P(+1) = 1e-3."""
    s = rng.uniform(-1, 1, (n, 13)).astype(np.float32)
    s[:, 7] = np.where(rng.random(n) < 0.5, 1, -1)
    s[:, 8] = np.where(rng.random(n) < 0.5, 1, -1)
    s[:, 12] = rng.uniform(0, 0.2, n)
    a = rng.uniform(-1, 1, (n, 4)).astype(np.float32)
    a[:, 3] = np.where(rng.random(n) < 1e-3, 1, -1)
    return s, a


def init_params(rng):
    """Seeded-init networks with the reference's bounds (HIRL.py:26-37,111-121)."""
    def U(b, shape):
        return rng.uniform(-b, b, shape).astype(np.float32)

    def block(in_dim, out_dim, names):
        fa, la, fb, lb, fin = names
        return {fa + ".weight": U(math.sqrt(6 / in_dim), (256, in_dim)), fa + ".bias": U(1 / math.sqrt(in_dim), (256,)),
                la + ".weight": np.ones(256, np.float32), la + ".bias": np.zeros(256, np.float32),
                fb + ".weight": U(math.sqrt(6 / 256), (512, 256)), fb + ".bias": U(1 / 16, (512,)),
                lb + ".weight": np.ones(512, np.float32), lb + ".bias": np.zeros(512, np.float32),
                fin + ".weight": U(1 / math.sqrt(512), (out_dim, 512)), fin + ".bias": U(1 / math.sqrt(512), (out_dim,))}

    actor = block(13, 4, ("full1", "layernorm1", "full2", "layernorm2", "final"))
    bc = block(13, 4, ("full1", "layernorm1", "full2", "layernorm2", "final"))
    critic = block(17, 1, ("full1", "layernorm1", "full2", "layernorm2", "final1"))
    critic.update(block(17, 1, ("full3", "layernorm3", "full4", "layernorm4", "final2")))
    return actor, critic, bc
