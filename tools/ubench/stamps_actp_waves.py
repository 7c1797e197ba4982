"""[r5] Per-wave phase lengths of ONE steady-state iteration (i = 4) of the persistent bf16 acting kernel's tile loop, waves 0 / 5 / 9 / 15 of workgroup 3
(diagnostic build: make -C hirl4ucav_amd/csrc stamps; the stamps themselves wait for the scalar memory pipe and so stretch what they measure a little):
    python tools/ubench/stamps_actp_waves.py [rows]
X: role work (LayerNorm 1 on waves 8-15, noise on wave 0, observation tile) | product MFMAs | bias + partial statistics | wait at barrier A
Y: layer-1 MFMAs | combine + LayerNorm 2 + final MFMA + store | last step (waves 0, 1) + pre-activation store | wait at barrier B        (x 10 ns)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from hirl4ucav_amd import _lib  # noqa: E402

_lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), "libhx_mi355_stamps.so")
from hirl4ucav_amd.agents import engine as E  # noqa: E402
from tests import _hirl_data as D  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
pp = D.make_params(1)
e = E.HirlEngine(batch=128)
e.load_params(pp["actor"], pp["critic"], pp["bc_actor"])
e.set_act_dtype("bf16")
obs = torch.rand((n, 13), device="cuda") * 2 - 1
out = torch.zeros((n, 4), device="cuda")
L = _lib.load()
w = np.zeros(80, np.float32)
names = ["X role work", "X product", "X bias+partial", "X wait A", "Y layer-1 MFMA", "Y combine+LN2+final MFMA", "Y last step+store", "Y wait B"]
for rep in range(3):
    for _ in range(3):
        e.act(obs, sigma=0.1, seed=3, out=out)
    torch.cuda.synchronize()
    assert L.hx_debug_stamps_actp(w.ctypes.data_as(ctypes.c_void_p)) == 0
    for slot, wave in enumerate((0, 5, 9, 15)):
        v = w[16 + 8 * slot:24 + 8 * slot]
        print(f"rep {rep} wave {wave:2d}: " + " | ".join(f"{nm} {int(x)}" for nm, x in zip(names, v)) + f" | sum {int(v.sum())} (x 10 ns)")
