"""[r5] act + env + insert (one launch, per-tile kernel) between 4,096 and 8,192 rows in the fp32 formats: the exact-split format in 16- / 32-row workgroups
(HX_ACT_X9_NRT2_ROWS) against fp32 MFMA from the fp32 image (x9_rows = None).  One process per variant (the knob is read once).
    python tools/ubench/act_x9_tiling_time.py"""
import os
import subprocess
import sys

CHILD = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from hirl4ucav_amd.agents import engine as E
from hirl4ucav_amd.environments.batched import BatchedHarfangEnv
from hirl4ucav_amd.utils.buffer import DeviceReplay
from tests import _hirl_data as D
pp = D.make_params(1)
e = E.HirlEngine(batch=128)
e.load_params(pp["actor"], pp["critic"], pp["bc_actor"])
if os.environ.get("FMT") == "f32i":
    e.x9_rows = None
if os.environ.get("FMT") == "bf16":
    e.set_act_dtype("bf16")
res = []
for n in (4096, 5120, 6144, 8192):
    rep = DeviceReplay(1 << 21, "cuda")
    env = BatchedHarfangEnv(n, scenario="circular", seed=5, max_step=1500, auto_reset=True, random_reset=True, replay=rep)
    env.reset()
    out = torch.zeros((n, 4), device="cuda")
    f = lambda: e.act_step(env, sigma=0.1, seed=3, out=out)
    for _ in range(20): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200): f()
    b.record(); torch.cuda.synchronize()
    res.append(f"{n}: {a.elapsed_time(b) * 1e3 / 200:6.2f}")
print(" | ".join(res))
'''
for name, env in (("x9, 16-row workgroups            ", {"FMT": "x9", "HX_ACT_X9_NRT2_ROWS": str(1 << 30)}),
                  ("x9, 32-row workgroups from 4,097 ", {"FMT": "x9", "HX_ACT_X9_NRT2_ROWS": "4097"}),
                  ("x9, 32-row workgroups from 8,192 ", {"FMT": "x9", "HX_ACT_X9_NRT2_ROWS": "8192"}),
                  ("fp32 MFMA from the fp32 image    ", {"FMT": "f32i"}),
                  ("bf16, 32-row workgroups from 8,192", {"FMT": "bf16"}),
                  ("bf16, 32-row workgroups from 4,097", {"FMT": "bf16", "HX_ACT_BF16_NRT2_ROWS": "4097"})):
    for rep in range(2):
        out = subprocess.run([sys.executable, "-c", CHILD], env={**os.environ, **env}, capture_output=True, text=True).stdout.strip().splitlines()
        print(name, "act + env + insert, us per launch ->", out[-1] if out else "(no output)", flush=True)
