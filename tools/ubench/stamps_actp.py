"""Phase stamps of one workgroup of the persistent acting kernel (csrc/hx_actp.hip), diagnostic build:
    make -C hirl4ucav_amd/csrc stamps && python tools/ubench/stamps_actp.py [dtype] [rows]
Words: setup | tiles 0, 1 | iteration 2: product + layer 1 | z2 -> LDS | barrier A | head (waves 0-7) or LN1 (8-15) | obs -> LDS | barrier B | rest of the loop | env tail   (x 10 ns)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from hirl4ucav_amd import _lib  # noqa: E402

_lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), "libhx_mi355_stamps.so")
from hirl4ucav_amd.agents import engine as E  # noqa: E402
from hirl4ucav_amd.environments.batched import BatchedHarfangEnv  # noqa: E402
from hirl4ucav_amd.utils.buffer import DeviceReplay  # noqa: E402
from tests import _hirl_data as D  # noqa: E402

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
pp = D.make_params(1)
e = E.HirlEngine(batch=128)
e.load_params(pp["actor"], pp["critic"], pp["bc_actor"])
if dt != "f32":
    e.set_act_dtype(dt)
rep = DeviceReplay(1 << 22, "cuda")
env = BatchedHarfangEnv(n, scenario=np.sort(np.arange(n) % 3).astype(np.int32), seed=5, max_step=1500, replay=rep)
env.reset()
out = torch.zeros((n, 4), device="cuda")
L = _lib.load()
w = np.zeros(80, np.float32)
for name, f in (("act", lambda: e.act(env.obs, sigma=0.1, seed=3, out=out)), ("act+env", lambda: e.act_step(env, sigma=0.1, seed=3, out=out))):
    for i in range(3):
        f()
        torch.cuda.synchronize()
        assert L.hx_debug_stamps_actp(w.ctypes.data_as(ctypes.c_void_p)) == 0
        k = int(w[0])
        print(dt, n, name, "x10ns:", " | ".join("%d" % v for v in w[1:k]))
