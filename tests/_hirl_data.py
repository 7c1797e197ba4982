"""Seeded inputs shared by the HIRL golden generator, the oracle tests and the GPU parity tests.  Everything is a pure
function of a seed (numpy Generator PCG64), so fixtures only need to carry indices, noise and outputs."""
import hashlib
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from oracle import hirl_oracle as H  # noqa: E402  (tests may use the oracle; the product never does)

PARAM_SEED, DATA_SEED = 2024, 77
N_REPLAY, N_EXPERT, N_EXPERT_ROWS = 2000, 600, 400


def make_params(seed, h1=256, h2=512):
    rng = np.random.default_rng(seed)
    return {"actor": H.init_actor(rng, h1=h1, h2=h2), "critic": H.init_critic(rng, h1=h1, h2=h2),
            "bc_actor": H.init_actor(rng, h1=h1, h2=h2)}


def plain_layernorm(params):
    """the same networks with every LayerNorm module at (1, 0): what a layerNorm=False agent holds (the reference constructs the modules
    either way and never uses or trains them in that mode, HIRL.py:28,33,114,119)"""
    out = {}
    for net, p in params.items():
        out[net] = {k: (np.ones_like(v) if k.endswith(".weight") else np.zeros_like(v)) if k.startswith("layernorm") else v for k, v in p.items()}
    return out


def _obs(rng, n):
    s = rng.uniform(-1, 1, (n, 13)).astype(np.float32)
    s[:, 7] = np.where(rng.random(n) < 0.5, 1, -1)
    s[:, 8] = np.where(rng.random(n) < 0.7, 1, -1)
    s[:, 12] = rng.uniform(0, 0.2, n)
    return s


def _rows(rng, n):
    rows = np.zeros((n, 32), np.float32)
    rows[:, 0:13] = _obs(rng, n)
    rows[:, 13:17] = rng.uniform(-1, 1, (n, 4))
    rows[:, 17:30] = _obs(rng, n)
    # rewards stay in the per-step range (distance/angle/fire penalties).  The +600 kill bonus is deliberately absent
    # from the free-running golden traces: one row with a 100x TD error makes a single ReLU sign flip (fp32 rounding)
    # visible as a ~1 % change of whole gradient tensors, and the traces then separate chaotically.  Outlier rows are
    # exercised by the per-call tests instead (tests/test_hirl_gpu.py::test_learn_ragged_batches).
    rows[:, 30] = rng.uniform(-9, 0, n) - np.where(rng.random(n) < 0.1, 8, 0)
    rows[:, 31] = rng.random(n) < 0.05
    return rows


def make_data(seed, outliers=False):
    rng = np.random.default_rng(seed)
    d = {"replay": _rows(rng, N_REPLAY), "expert_rows": _rows(rng, N_EXPERT_ROWS), "expert_s": _obs(rng, N_EXPERT)}
    a = rng.uniform(-1, 1, (N_EXPERT, 4)).astype(np.float32)
    a[:, 3] = np.where(rng.random(N_EXPERT) < 0.05, 1, -1)
    d["expert_a"] = a
    if outliers:  # kill bonus rows (HarfangEnv_GYM.py:135-136)
        k = rng.random(N_REPLAY) < 0.02
        d["replay"][k, 30] += 600
    return d


def probe_index(size, count=128):
    return (np.arange(count, dtype=np.int64) * 2654435761 % size).astype(np.int64)


def checksum(tree):
    h = hashlib.sha256()
    for k in sorted(tree):
        v = tree[k]
        if isinstance(v, dict):
            h.update(checksum(v).encode())
        else:
            h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()


def net_probe(flat):
    flat = np.asarray(flat, np.float64)
    return flat.sum(), np.abs(flat).sum(), flat[probe_index(flat.size)]
