"""Env-step kernel (fused insert) duration by launch shape AND by the age of the state: the same launch costs more once missiles fly and
locks hold (more waves take the missile / lock branches), so a sweep's verdict on a layout depends on how many steps preceded the stamps.
    python tools/ubench/env_layout_by_age.py [--sizes 262144,1048576,4194304] [--layouts s256,p256] [--ages 3,40,200]
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
p.add_argument("--sizes", default="262144,1048576,4194304")
p.add_argument("--layouts", default="s256,p256")
p.add_argument("--ages", default="3,40,200")
p.add_argument("--scenario", default="straight_line")
args = p.parse_args()
for n in (int(x) for x in args.sizes.split(",")):
    for tag in args.layouts.split(","):
        rep = DeviceReplay(max(2 * n, 1 << 20))
        env = BatchedHarfangEnv(n, scenario=args.scenario, seed=0, max_step=1500, replay=rep,
                                layout=0 if tag == "auto" else _lib.layout(tag[0] == "p", int(tag[1:])))
        env.reset()
        a = torch.rand(n, 4, device="cuda") * 2 - 1
        done = 0
        for age in (int(x) for x in args.ages.split(",")):
            while done < age:
                env.step(a)
                done += 1
            us = float(np.median(stamped_env_us(env, a, 16)))
            done += 16
            st = env.stats_dict()
            print(json.dumps({"n": n, "layout": tag, "steps_before": age, "kernel_us": round(us, 2), "frac_of_8TBps": round(550 * n / us / 1e3 / 8000, 4),
                              "missile_fires": st.get("missile_fires"), "locked_steps": st.get("locked_steps")}), flush=True)
        del env, rep
        torch.cuda.empty_cache()
