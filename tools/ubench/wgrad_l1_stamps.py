"""Phase stamps (x 10 ns) of ONE layer-1 workgroup (block 96, job 0), one tile workgroup (block 0) and one vector workgroup (block 64) of the critics' wgrad + Adam
launch in STEADY STATE (the last launch of a running bench loop; tools/ubench/stamps.py times an isolated call).  Stamps build.
    python3 tools/ubench/wgrad_l1_stamps.py [bench flags]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from hirl4ucav_amd import _lib  # noqa: E402

_lib.SO_PATH = os.environ.get("HX_STAMPS_LIB") or os.path.join(os.path.dirname(_lib.SO_PATH), "libhx_mi355_stamps.so")
import bench as B  # noqa: E402

loop = B.Loop(B.parse(sys.argv[1:]), 0, 1, torch.device("cuda", 0))
L = _lib.load()
out = np.zeros(80, np.float32)
for rep in range(4):
    for _ in range(41 if rep % 2 else 40):  # end on a critic-only / an actor step alternately
        loop.step()
    torch.cuda.synchronize()
    assert L.hx_debug_stamps(out.ctypes.data_as(ctypes.c_void_p)) == 0
    n_t, n_v, n_l = int(out[32]), int(out[40]), int(out[48])
    print("last wgrad launch (%s step):" % ("actor" if not loop.eng.actor_trainable else "critic-only"))
    print("  tile wg   (%2d stamps):" % n_t, [int(v) for v in out[33:32 + n_t]], "sum", int(out[33:32 + n_t].sum()))
    print("  vector wg (%2d stamps):" % n_v, [int(v) for v in out[41:40 + n_v]], "sum", int(out[41:40 + n_v].sum()))
    print("  layer-1   (%2d stamps):" % n_l, [int(v) for v in out[49:48 + n_l]], "sum", int(out[49:48 + n_l].sum()))
