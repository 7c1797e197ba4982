"""Workload for the PMC passes over the persistent acting kernel (tools/pmc_actp_passes.sh): 40 launches of act + env at one size."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from hirl4ucav_amd.agents import engine as E  # noqa: E402
from hirl4ucav_amd.environments.batched import BatchedHarfangEnv  # noqa: E402
from hirl4ucav_amd.utils.buffer import DeviceReplay  # noqa: E402
from tests import _hirl_data as D  # noqa: E402

dt = os.environ.get("ACT_DTYPE", "bf16")
n = int(os.environ.get("ACT_ROWS", "131072"))
pp = D.make_params(1)
e = E.HirlEngine(batch=128)
e.load_params(pp["actor"], pp["critic"], pp["bc_actor"])
if dt != "f32":
    e.set_act_dtype(dt)
rep = DeviceReplay(1 << 22, "cuda")
env = BatchedHarfangEnv(n, scenario=np.sort(np.arange(n) % 3).astype(np.int32), seed=5, max_step=1500, replay=rep)
env.reset()
out = torch.zeros((n, 4), device="cuda")
for _ in range(40):
    e.act(env.obs, sigma=0.1, seed=3, out=out)
for _ in range(40):
    e.act_step(env, sigma=0.1, seed=3, out=out)
torch.cuda.synchronize()
print("done", dt, n)
