"""Rate of the single-env compatibility facade (host numpy in / out every call: the PCIe- and sync-inclusive path) next to the
reference's socket env it replaces.  HarfangEnv.step + Agent.chooseAction + store + learn, as train_all.py:343-361 calls them."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from hirl4ucav_amd.agents.HIRL import Agent as HIRLAgent  # noqa: E402
from hirl4ucav_amd.environments.HarfangEnv_GYM import HarfangEnv  # noqa: E402

env = HarfangEnv()
rng = np.random.default_rng(0)
es, ea = rng.uniform(-1, 1, (2000, 13)), rng.uniform(-1, 1, (2000, 4))
agent = HIRLAgent(1e-3, 1e-3, 13, 4, 256, 512, 0.005, 0.99, 100000, 128, True, "Harfang_GYM", es, ea, 0.5, True)
state = env.reset()
for _ in range(300):  # fill the buffer
    a = env.action_space.sample()
    n_state, r, d, _, s = env.step(a)
    agent.store(state, a, n_state, r, d, s)
    state = n_state
K = 1000
t0 = time.perf_counter()
for _ in range(K):
    n_state, r, d, _, s = env.step(env.action_space.sample())
t1 = time.perf_counter()
for _ in range(K):
    a = agent.chooseAction(state)
    n_state, r, d, _, s = env.step(a)
    agent.store(state, a, n_state, r, d, s)
    agent.learn(100, 0, 0.0)
    state = n_state
t2 = time.perf_counter()
print(f"facade env.step alone: {K / (t1 - t0):,.0f} steps/s; chooseAction + step + store + learn: {K / (t2 - t1):,.0f} steps/s", flush=True)
