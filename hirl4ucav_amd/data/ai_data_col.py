#!/usr/bin/env python3
"""Expert-data collector — counterpart of hirl/data/<scenario>/ai_data_col.py for the build's own simulator.

The reference flies 20 episodes with Harfang's IA autopilot, records (observation, control read-back) per tick until the
opponent is destroyed, drops episodes that took more than 2000 steps or whose launch came more than 20 steps after (or
before) the first lock (ai_data_col.py:39-94), and writes the two-row CSV read by utils/data_processor.read_data
(ai_data_col.py:101-111).  Here all episodes fly at once, one env each, with the scripted pilot of expert_pilot.py.

    python -m hirl4ucav_amd.data.ai_data_col --env straight_line --random --episodes 20 --out expert_data/straight_line/expert_data_ai_random.csv
"""
import argparse
import os

import numpy as np
import torch

from ..environments.batched import SCENARIOS, BatchedHarfangEnv
from ..utils.data_processor import write_data
from .expert_pilot import pursuit_actions

MAX_EPISODE_STEP = 2000  # ai_data_col.py:86


def collect(scenario, episodes, if_random, seed=0, noise_std=0.0, device="cuda"):
    """-> (states [M, 13], actions [M, 4], info): the valid episodes back to back, each from its reset observation to the
    observation that shows the opponent destroyed (the IA loop records that last state too, ai_data_col.py:57-60)."""
    env = BatchedHarfangEnv(episodes, scenario=scenario, device=device, seed=seed, auto_reset=False, random_reset=if_random, collect_stats=False)
    gen = torch.Generator(device=device)
    gen.manual_seed(seed)
    obs = env.reset()
    S = torch.zeros((MAX_EPISODE_STEP + 1, episodes, 13), device=device)
    A = torch.zeros((MAX_EPISODE_STEP + 1, episodes, 4), device=device)
    steps = 0
    for t in range(MAX_EPISODE_STEP + 1):
        a = pursuit_actions(env.state, obs, noise_std, gen)
        S[t], A[t] = obs, a
        steps = t + 1
        if t % 64 == 63 and bool((obs[:, 12] <= 0).all()):
            break
        obs = env.step(a)[0]
    return filter_episodes(S[:steps].cpu().numpy().astype(np.float64), A[:steps].cpu().numpy().astype(np.float64))


def filter_episodes(S, A):
    """S [T, E, 13], A [T, E, 4] (one column per episode) -> (states, actions, info): the reference's validity rule
    (ai_data_col.py:86): an episode counts when the opponent was destroyed within 2000 steps and the launch came 0..20 steps
    after the first tick with lock and missile; each kept episode runs from its reset observation to the first observation
    that shows the opponent destroyed."""
    steps, episodes = S.shape[0], S.shape[1]
    states, actions, delt, invalid, lengths = [], [], [], 0, []
    for e in range(episodes):
        dead = np.nonzero(S[:, e, 12] <= 0)[0]
        n = int(dead[0]) + 1 if len(dead) else steps  # rows 0 .. first destroyed observation
        can = np.nonzero((S[:n, e, 7] > 0) & (S[:n, e, 8] > 0))[0]
        fired = np.nonzero(A[:n, e, 3] > 0)[0]
        lock, fire = (int(can[0]) if len(can) else 0), (int(fired[0]) if len(fired) else 0)
        if not len(dead) or n > MAX_EPISODE_STEP or fire - lock > 20 or fire - lock < 0:
            invalid += 1
            continue
        states.append(S[:n, e]); actions.append(A[:n, e]); delt.append(fire - lock); lengths.append(n)
    info = {"episodes": episodes, "invalid": invalid, "lengths": lengths, "delt": delt}
    if not states:
        return np.zeros((0, 13)), np.zeros((0, 4)), info
    return np.concatenate(states), np.concatenate(actions), info


def main(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--env", default="straight_line", choices=list(SCENARIOS))
    p.add_argument("--random", action="store_true")
    p.add_argument("--episodes", type=int, default=20)  # ai_data_col.py:38
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--noise", type=float, default=0.0, help="std of Gaussian noise on the pilot's stick levels")
    p.add_argument("--out", default=None)
    c = p.parse_args(argv)
    out = c.out or os.path.join("expert_data", c.env, "expert_data_ai_random.csv" if c.random else "expert_data_ai.csv")  # ai_data_col.py:104-105
    s, a, info = collect(c.env, c.episodes, c.random, c.seed, c.noise)
    os.makedirs(os.path.dirname(out) or ".", exist_ok=True)
    write_data(s, a, out)
    print(f"episode {info['episodes']}  step {len(s)}  invalid data {info['invalid']}")
    print(info["delt"])
    print("Finish", out)
    return out


if __name__ == "__main__":
    main()
