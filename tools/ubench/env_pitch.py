"""Env-step kernel (fused insert) against the PITCH of the struct-of-arrays state (floats between two state words' arrays; the C ABI's `stride`).
With pitch = n = a power of two, the ~45 arrays one wave reads back to back are a power of two apart.
    python tools/ubench/env_pitch.py [--sizes 65536,1048576,4194304] [--pads 0,64,1056,16448] [--layouts auto]
"""
import argparse
import json
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from bench import stamped_env_us  # noqa: E402
from hirl4ucav_amd import _lib  # noqa: E402
from hirl4ucav_amd.environments.batched import BatchedHarfangEnv  # noqa: E402
from hirl4ucav_amd.utils.buffer import DeviceReplay  # noqa: E402

p = argparse.ArgumentParser()
p.add_argument("--sizes", default="65536,1048576,4194304")
p.add_argument("--pads", default="0,64,1056,16448,0")
p.add_argument("--layouts", default="auto")
args = p.parse_args()
for n in (int(x) for x in args.sizes.split(",")):
    for tag in args.layouts.split(","):
        for pad in (int(x) for x in args.pads.split(",")):
            rep = DeviceReplay(max(2 * n, 1 << 20))
            env = BatchedHarfangEnv(n, scenario="straight_line", seed=0, max_step=1500, replay=rep, pitch=n + pad,
                                    layout=0 if tag == "auto" else _lib.layout(tag[0] == "p", int(tag[1:])))
            env.reset()
            a = torch.rand(n, 4, device="cuda") * 2 - 1
            for _ in range(3):
                env.step(a)
            us = stamped_env_us(env, a, 24)
            print(json.dumps({"n": n, "layout": tag, "pitch_pad_floats": pad, "kernel_us": round(float(np.median(us)), 2), "min": round(float(np.min(us)), 2),
                              "frac_of_8TBps": round(550 * n / float(np.median(us)) / 1e3 / 8000, 4), "state_ptr_mod_2MB": env.state.data_ptr() % (2 << 20)}), flush=True)
            del env, rep
            torch.cuda.empty_cache()
