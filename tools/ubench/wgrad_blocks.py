"""Which workgroups of the critics' wgrad + Adam launch are its longest?  Stamps build; life span of every workgroup by blockIdx.x
(0..63 dW2 tiles, 64..95 the 512-wide vector gradients, 96..111 layer 1) of a few steady-state launches of bench.py's loop.
    python3 tools/ubench/wgrad_blocks.py [bench flags]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools/ubench")
from hirl4ucav_amd import _lib  # noqa: E402

_lib.SO_PATH = os.environ.get("HX_STAMPS_LIB") or os.path.join(os.path.dirname(_lib.SO_PATH), "libhx_mi355_stamps.so")
import bench as B  # noqa: E402
from spans import fetch_blocks  # noqa: E402

loop = B.Loop(B.parse(sys.argv[1:]), 0, 1, torch.device("cuda", 0))
L = _lib.load()
L.hx_debug_spans.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint)]
for _ in range(40):
    loop.step()
torch.cuda.synchronize()
fetch_blocks(L)
for _ in range(6):
    loop.step()
torch.cuda.synchronize()
spans, tags, bx, by = fetch_blocks(L)
for tag, name in ((4, "wgrad"), (3, "bwd_l2"), (1, "fwd_l2"), (5, "front A"), (6, "front B")):
    m = tags == tag
    if not m.any():
        continue
    s, x, y = spans[m], bx[m], by[m]
    # launches: cluster by start time (gaps > 3 us)
    order = np.argsort(s[:, 0])
    s, x, y = s[order], x[order], y[order]
    cuts = [0] + [i for i in range(1, len(s)) if s[i, 0] - s[i - 1, 0] > 300] + [len(s)]
    for a, b in list(zip(cuts[:-1], cuts[1:]))[-3:]:
        t0 = s[a:b, 0].min()
        dur = (s[a:b, 1] - s[a:b, 0]) / 100.0
        end = (s[a:b, 1] - t0) / 100.0
        print(f"{name}: launch of {b - a} workgroups, last exit {end.max():.2f} us after the first start")
        xs = x[a:b]
        for lo, hi, what in ((0, 64, "tiles / blocks 0..63"), (64, 96, "blocks 64..95"), (96, 112, "blocks 96..111"), (112, 1 << 16, "blocks 112..")):
            mm = (xs >= lo) & (xs < hi)
            if mm.any():
                print(f"    {what:22s} n={mm.sum():3d}  duration mean {dur[mm].mean():5.2f} max {dur[mm].max():5.2f}   exit mean {end[mm].mean():5.2f} max {end[mm].max():5.2f}")
