"""Whole-run snapshots for true resume.

The reference only writes the four network state_dicts (HIRL.py:336-350), so a restarted run loses the Adam moments,
the replay contents, the update counters and the episode progress.  A snapshot here holds everything the next vector
step reads: the engine arena (all networks, gradients, Adam moments, soft weight), the engine's call counters (they key
the Philox streams of the sampler and of the exploration noise), the env state words + episode counters + statistics,
the filled part of the replay ring, the host RNGs, and the driver's own scalars.  A run resumed from a snapshot
continues bit-identically to the run that was never stopped (tests/test_facade_gpu.py).
"""
import random

import numpy as np
import torch

ENGINE_COUNTERS = ("critic_step", "actor_step", "update_count", "actor_trainable", "sample_calls", "act_calls", "learning_steps")


def engine_state(eng):
    st = {"arena": eng.arena.detach().cpu().clone(), "counters": {k: getattr(eng, k) for k in ENGINE_COUNTERS if hasattr(eng, k)}}
    if hasattr(eng, "acting_format"):
        st["acting_format"] = eng.acting_format()  # the run's arithmetic is part of its state (train_all.py: --dtype; here: what the flag does not say)
    return st


def load_engine_state(eng, st):
    if st["arena"].numel() != eng.arena.numel():
        raise ValueError(f"snapshot arena has {st['arena'].numel()} words, this engine {eng.arena.numel()} (different agent or batch size)")
    was, now = st.get("acting_format"), (eng.acting_format() if hasattr(eng, "acting_format") else None)
    if now is not None and was != now:
        import warnings

        # (a snapshot from before round 6 carries no format: round 5's default, or round 4's x9_rows = 16,384 with nine terms — it cannot tell)
        warnings.warn(f"snapshot was written under acting format {was if was is not None else 'unrecorded (a round <= 5 snapshot)'}, this engine "
                      f"acts under {now}: the run continues under other acting arithmetic (fp32 results up to summation order)", stacklevel=2)
    eng.arena.copy_(st["arena"])
    if hasattr(eng, "needs_reload"):
        eng.needs_reload = False  # host counters and device state agree again from here
    if hasattr(eng, "front_reset"):
        eng.front_reset()
    if hasattr(eng, "refresh_bf16"):
        eng.refresh_bf16()  # the bf16 image of the actor's W2 follows the restored fp32 parameters
    for k, v in st["counters"].items():
        setattr(eng, k, v)


def replay_state(replay):
    n = min(int(replay.total.item()), replay.capacity)
    return {"capacity": replay.capacity, "total": int(replay.total.item()), "ring": replay.ring[:n].cpu().clone(),
            "success": replay.success[:n].cpu().clone()}


def load_replay_state(replay, st):
    if st["capacity"] != replay.capacity:
        raise ValueError(f"snapshot replay capacity {st['capacity']} != {replay.capacity}")
    n = st["ring"].shape[0]
    replay.ring[:n].copy_(st["ring"])
    replay.success[:n].copy_(st["success"])
    replay.total.fill_(st["total"])


def env_state(env):
    return {"state": env.state.cpu().clone(), "obs": env.obs.cpu().clone(), "episode_ctr": env.episode_ctr.cpu().clone(),
            "stats": None if env.stats is None else env.stats.cpu().clone()}


def load_env_state(env, st):
    if tuple(st["state"].shape) != tuple(env.state.shape):
        raise ValueError(f"snapshot holds {st['state'].shape[1]} envs, this run {env.n}")
    env.state.copy_(st["state"])
    env.obs.copy_(st["obs"])
    env.episode_ctr.copy_(st["episode_ctr"])
    if env.stats is not None and st["stats"] is not None:
        if tuple(st["stats"].shape) == tuple(env.stats.shape):
            env.stats.copy_(st["stats"])
        else:  # a snapshot from before the counters were kept HX_STAT_WAYS times (and possibly with fewer statistics): its totals go into way 0
            old = st["stats"].reshape(-1)
            k = min(env.stats.shape[1], old.numel())
            env.stats.zero_()
            env.stats[0, :k].copy_(old[:k])


def save_run(path, eng, env, replay, driver):
    """Written to a temporary file and renamed: a kill during the ~130 MB write leaves the previous snapshot intact."""
    import os

    torch.cuda.synchronize()
    tmp = path + ".tmp"
    torch.save({"engine": engine_state(eng), "env": env_state(env), "replay": replay_state(replay), "driver": dict(driver),
                "rng": {"python": random.getstate(), "numpy": np.random.get_state(), "torch": torch.get_rng_state(),
                        "torch_cuda": torch.cuda.get_rng_state(env.device)}}, tmp)
    os.replace(tmp, path)


def read_run(path):
    """the snapshot as stored (its driver dict carries the run's seed: the caller needs it BEFORE it builds the env)"""
    return torch.load(path, map_location="cpu", weights_only=False)


def load_run(path, eng, env, replay):
    snap = path if isinstance(path, dict) else read_run(path)
    load_engine_state(eng, snap["engine"])
    load_env_state(env, snap["env"])
    load_replay_state(replay, snap["replay"])
    random.setstate(snap["rng"]["python"])
    np.random.set_state(snap["rng"]["numpy"])
    torch.set_rng_state(snap["rng"]["torch"])
    torch.cuda.set_rng_state(snap["rng"]["torch_cuda"], env.device)
    return snap["driver"]
