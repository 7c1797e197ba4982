"""Sweep of the env-step kernel (K1) over sizes and launch shapes: the kernel's OWN duration (begin/end stamps of
hipExtLaunchKernelGGL, what rocprofv3 reports) and the achieved algorithmic GB/s vs the 8 TB/s HBM peak.
Algorithmic bytes: 370 B/env-step kernel only, 550 B/env-step with the fused replay insert.

    python tools/sweep_env_kernel.py [--sizes 4096,65536,...] [--layouts auto,p32,p64,...,s256] [--no-insert]
"""
import argparse
import ctypes
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from hirl4ucav_amd import _lib  # noqa: E402
from hirl4ucav_amd.environments.batched import BatchedHarfangEnv  # noqa: E402
from hirl4ucav_amd.utils.buffer import DeviceReplay  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--sizes", default="4096,16384,65536,262144,1048576,4194304")
p.add_argument("--layouts", default="auto,p32,p64,p128,p256,p512,s64,s128,s256")
p.add_argument("--no-insert", action="store_true")
p.add_argument("--iters", type=int, default=24)
args = p.parse_args()

L = _lib.load()


def lay(tag):
    return 0 if tag == "auto" else _lib.layout(tag[0] == "p", int(tag[1:]))


out = []
for insert in ([False] if args.no_insert else [True, False]):
    for n in (int(x) for x in args.sizes.split(",")):
        for tag in args.layouts.split(","):
            rep = DeviceReplay(max(2 * n, 1 << 20)) if insert else None
            env = BatchedHarfangEnv(n, scenario="straight_line", seed=0, max_step=1500, auto_reset=True, replay=rep, layout=lay(tag))
            env.reset()
            a = torch.rand(n, 4, device="cuda") * 2 - 1
            for _ in range(5):
                env.step(a)
            evs = [(ctypes.c_void_p(L.hx_event_create()), ctypes.c_void_p(L.hx_event_create())) for _ in range(args.iters)]
            torch.cuda.synchronize()
            for s, e in evs:
                env.time_next_steps(s, e)
                env.step(a)
            env.time_next_steps(None, None)
            torch.cuda.synchronize()
            us = []
            for s, e in evs:
                v = ctypes.c_float()
                _lib.call("hx_event_elapsed_us", s, e, ctypes.byref(v))
                us.append(v.value)
                L.hx_event_destroy(s)
                L.hx_event_destroy(e)
            # back-to-back un-stamped launches: what a loop sees per step (kernel + launch boundary)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                env.step(a)
            e1.record()
            torch.cuda.synchronize()
            b = 550 if insert else 370
            k = float(np.median(us))
            row = {"n": n, "insert": insert, "layout": tag, "kernel_us": round(k, 2), "kernel_us_min": round(float(np.min(us)), 2),
                   "stream_us": round(e0.elapsed_time(e1) * 1e3 / args.iters, 2), "GBps": round(b * n / k / 1e3, 1),
                   "frac_of_8TBps": round(b * n / k / 1e3 / 8000, 4)}
            print(json.dumps(row), flush=True)
            out.append(row)
            del env, rep
            torch.cuda.empty_cache()
