"""Acting launch at large sizes: persistent kernel (hx_actp.hip) against the per-tile kernel (HX_ACT_PERSIST=0), act alone and act + env
+ insert, per dtype.  torch events around 50 back-to-back launches (launch boundary included).
  python tools/ubench/actp_time.py [dtype ...]      dtype in bf16 f32 f32x9 sac"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from hirl4ucav_amd.agents import engine as E  # noqa: E402
from hirl4ucav_amd.environments.batched import BatchedHarfangEnv  # noqa: E402
from hirl4ucav_amd.utils.buffer import DeviceReplay  # noqa: E402
from tests import _hirl_data as D  # noqa: E402


def t(f, n=50):
    for _ in range(10):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def both(f):
    os.environ.pop("HX_ACT_PERSIST", None)
    p = t(f)
    os.environ["HX_ACT_PERSIST"] = "0"
    q = t(f)
    os.environ.pop("HX_ACT_PERSIST", None)
    return p, q


dtypes = sys.argv[1:] or ["bf16", "f32"]
sizes = [int(s) for s in os.environ.get("SIZES", "16384,65536,131072").split(",")]
pp = D.make_params(1)
for dt in dtypes:
    if dt == "sac":
        from hirl4ucav_amd.agents import sac_engine as SE
        from tests.test_oracle_sac import sac_params
        p = sac_params()
        e = SE.SacEngine(batch=128)
        e.load_params(p["policy"], p["q1"], p["q2"])
    else:
        e = E.HirlEngine(batch=128)
        e.load_params(pp["actor"], pp["critic"], pp["bc_actor"])
        if dt != "f32":
            e.set_act_dtype(dt)
    for n in sizes:
        scen = np.sort(np.arange(n) % 3).astype(np.int32)
        rep = DeviceReplay(1 << 22, "cuda")
        env = BatchedHarfangEnv(n, scenario=scen, seed=5, max_step=1500, auto_reset=True, random_reset=True, replay=rep)
        env.reset()
        out = torch.zeros((n, 4), device="cuda")
        if dt == "sac":
            a = both(lambda: e.act(env.obs, seed=3, out=out))
            s = both(lambda: e.act_step(env, seed=3, out=out))
        else:
            a = both(lambda: e.act(env.obs, sigma=0.1, seed=3, out=out))
            s = both(lambda: e.act_step(env, sigma=0.1, seed=3, out=out))
        en = t(lambda: env.step(out))
        print(f"{dt:6s} n={n:7d}  act: persistent {a[0]:7.1f} us | per-tile {a[1]:7.1f} us    act+env: one launch {s[0]:7.1f} us | per-tile path {s[1]:7.1f} us"
              f"    env alone {en:6.1f} us", flush=True)
