"""Workload for the HBM-traffic counters of the env-step kernel (run under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one
counter per pass): a calibration copy with the same access shape over a known byte count, then env steps at HX_PMC_ENVS envs
(default 1,048,576); HX_PMC_FUSED=1 adds six launches of the fused act + env kernel at that size (HX_PMC_DTYPE f32 | bf16: the policy's format),
HX_PMC_FRONT=1 eight front launches (HirlEngine.step_learn).  tools/pmc_traffic_json.py turns
the passes' CSV files into profiles/pmc_env_traffic.json."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hirl4ucav_amd import _lib
from hirl4ucav_amd.environments.batched import BatchedHarfangEnv
from hirl4ucav_amd.utils.buffer import DeviceReplay
import ctypes
_lib.register("hx_debug_copy_dword", [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p])
n = int(os.environ.get("HX_PMC_ENVS", 1 << 20))
cal = 128 << 20  # floats: 512 MiB read + 512 MiB written, well past the 256 MiB Infinity Cache
a, b = torch.zeros(cal, device="cuda"), torch.zeros(cal, device="cuda")
for _ in range(3):
    _lib.call("hx_debug_copy_dword", a.data_ptr(), b.data_ptr(), cal, _lib.stream_ptr())
rep = DeviceReplay(max(4 * n, 1 << 20))
env = BatchedHarfangEnv(n, scenario="straight_line", seed=0, max_step=1500, auto_reset=True, replay=rep)
env.reset()
act = torch.rand(n, 4, device="cuda") * 2 - 1
for _ in range(6):
    env.step(act)
if os.environ.get("HX_PMC_FUSED"):  # the kernel of bench.py's timed loop: policy inference + env step + insert in ONE launch (act_fused_kernel<..., ENV>)
    from hirl4ucav_amd.agents.engine import HirlEngine
    eng = HirlEngine(batch=128)
    if os.environ.get("HX_PMC_DTYPE", "f32") == "bf16":  # (beyond 8,192 envs the launch is the persistent kernel of hx_actp.hip: bf16 weight-stationary,
        eng.set_act_dtype("bf16")                          #  fp32 through the exact 9-term split from 16,384 rows on)
    out = torch.empty((n, 4), device="cuda")
    for _ in range(6):
        eng.act_step(env, sigma=0.1, seed=1, out=out)
if os.environ.get("HX_PMC_FRONT"):  # bench.py's default loop where it applies: the FRONT launch (act_front_kernel: env step + launches A and B of learn())
    import numpy as np
    from hirl4ucav_amd.agents.engine import HirlEngine
    from tests import _hirl_data as D
    eng = HirlEngine(batch=128)
    if os.environ.get("HX_PMC_DTYPE", "f32") == "bf16":  # the bf16 update path with the bf16 acting format (persistent acting role beyond 4,096 envs)
        eng.set_update_dtype("bf16")
        eng.set_act_dtype("bf16")
    pp = D.make_params(1)
    eng.load_params(pp["actor"], pp["critic"], pp["bc_actor"])
    rng = np.random.default_rng(0)
    exp = DeviceReplay(64)
    exp.store_rows(torch.from_numpy(rng.normal(size=(40, 32)).astype(np.float32)))
    bc = torch.from_numpy(rng.normal(size=(150, 32)).astype(np.float32)).cuda()
    out = torch.empty((n, 4), device="cuda")
    for k in range(8):
        eng.step_learn(env, exp, bc, n_main=128, act_sigma=0.1, act_seed=1, out=out, sample_seed=2, bc_weight_now=100 if k == 0 else None)
    eng.front_check()
torch.cuda.synchronize()
print("done", n, cal)
