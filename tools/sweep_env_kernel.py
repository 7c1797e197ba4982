"""Size sweep of the env-step kernel (K1): achieved algorithmic GB/s vs the 8 TB/s HBM peak.
Algorithmic bytes: 370 B/env-step kernel only, 550 B/env-step with the fused replay insert."""
import json
import sys

import torch

sys.path.insert(0, ".")
from hirl4ucav_amd.environments.batched import BatchedHarfangEnv  # noqa: E402
from hirl4ucav_amd.utils.buffer import DeviceReplay  # noqa: E402

sizes = [4096, 16384, 65536, 262144, 1 << 20, 1 << 22]
out = []
for insert in (False, True):
    for n in sizes:
        rep = DeviceReplay(max(2 * n, 1 << 20)) if insert else None
        env = BatchedHarfangEnv(n, scenario="straight_line", seed=0, max_step=1500, auto_reset=True, replay=rep)
        env.reset()
        a = torch.rand(n, 4, device="cuda") * 2 - 1
        for _ in range(5):
            env.step(a)
        iters = 50 if n <= (1 << 20) else 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(iters):
            env.step(a)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        b = 550 if insert else 370
        row = {"n": n, "insert": insert, "us_per_launch": round(us, 2), "env_steps_per_s": round(n / us * 1e6),
               "GBps": round(b * n / us / 1e3, 1), "frac_of_8TBps": round(b * n / us / 1e3 / 8000, 4)}
        print(json.dumps(row), flush=True)
        out.append(row)
        del env, rep
        torch.cuda.empty_cache()
