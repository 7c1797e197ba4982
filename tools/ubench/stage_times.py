"""Back-to-back timing of each C-ABI stage (HIP events on the launch stream), to separate kernel time from
profiler/launch artefacts.  Run on the GPU box: python tools/ubench/stage_times.py"""
import sys, time
import torch
sys.path.insert(0, ".")
import bench as B
from hirl4ucav_amd import _lib
import ctypes

args = B.parse.__wrapped__() if hasattr(B.parse, "__wrapped__") else None
class A: envs=4096; batch=128; scenario="straight_line"
loop = B.Loop(A, 0, 1, torch.device("cuda", 0))
for _ in range(50): loop.step()
e, env = loop.eng, loop.env
def timeit(name, fn, n=300):
    for _ in range(20): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); host = (time.perf_counter() - t0) / n * 1e6; torch.cuda.synchronize()
    print(f"{name:34s} gpu {e0.elapsed_time(e1) * 1e3 / n:8.2f} us/call   host-issue {host:6.2f} us/call", flush=True)
idx, idx_bc, noise = e.sample(loop.replay, loop.expert, loop.bc_table, n_main=128, seed=2)
nets, hyper = ctypes.byref(e.nets), ctypes.byref(e.hyper)
from hirl4ucav_amd.agents.engine import HxBatch
batch = HxBatch(e.rows.data_ptr(), e.bc_rows.data_ptr(), 128, noise.data_ptr())
st = _lib.stream_ptr()
timeit("sample+gather", lambda: e.sample(loop.replay, loop.expert, loop.bc_table, n_main=128, seed=2))
timeit("env.step (4096, insert)", lambda: env.step(loop.actions))
timeit("act (fwd wide + head, 4096)", lambda: e.act(env.obs, sigma=0.1, seed=1, out=loop.actions))
timeit("critic_grads (A,B,C,D)", lambda: _lib.call("hx_hirl_critic_grads", nets, ctypes.byref(batch), hyper, 0, st))
timeit("adam critic", lambda: _lib.call("hx_adam", nets, hyper, 0, 5, 1.0, 0, 0.0, 0.0, 128, st))
timeit("actor_backward (F,G,H,I) soft", lambda: _lib.call("hx_hirl_actor_backward", nets, ctypes.byref(batch), hyper, 1, 0, st))
timeit("actor_backward (F,G,H,I) nosoft", lambda: _lib.call("hx_hirl_actor_backward", nets, ctypes.byref(batch), hyper, 0, 0, st))
timeit("actor_wgrad (J)", lambda: _lib.call("hx_hirl_actor_wgrad", nets, hyper, 128, 128, 2, 0.0, 0.0, st))
timeit("adam actor", lambda: _lib.call("hx_adam", nets, hyper, 1, 5, 1.0, 2, 0.0, 0.0, 128, st))
timeit("polyak", lambda: _lib.call("hx_polyak", nets, hyper, st))
timeit("hx_hirl_learn critic-only (4 k)", lambda: _lib.call("hx_hirl_learn", nets, ctypes.byref(batch), hyper, 5, 0, 5, 0, 2, 0.0, 0.0, st))
timeit("hx_hirl_learn actor call (8 k)", lambda: _lib.call("hx_hirl_learn", nets, ctypes.byref(batch), hyper, 5, 1, 5, 0, 2, 0.0, 0.0, st))
timeit("full loop.step()", lambda: loop.step(), n=500)
