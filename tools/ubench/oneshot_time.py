"""Duration of hx_allreduce_oneshot's kernel with world = 1 (no peers: announce + fences + copy of the 1.1 MB critic message) — what the
exchange costs a rank before any waiting; HX_LIBRARY selects the build."""
import ctypes
import sys

import torch

sys.path.insert(0, ".")
from hirl4ucav_amd import _lib  # noqa: E402
from hirl4ucav_amd.agents import engine as E  # noqa: E402,F401  (registers the entry points)

n = 276484 // 4 * 4
msg = ctypes.c_void_p()
flag = ctypes.c_void_p()
_lib.call("hx_ipc_alloc", n * 4, 0, ctypes.byref(msg))
_lib.call("hx_ipc_alloc", 256, 1, ctypes.byref(flag))
dst = torch.zeros(n, device="cuda")
bufs = (ctypes.c_void_p * 1)(msg)
flags = (ctypes.c_void_p * 1)(flag)
status = ctypes.c_void_p(flag.value + 128)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
epoch = 0
for rep in range(3):
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(500):
        epoch += 1
        _lib.call("hx_allreduce_oneshot", dst.data_ptr(), bufs, flags, status, 1, 0, n, epoch, 1000, _lib.stream_ptr())
    ev1.record()
    torch.cuda.synchronize()
    print("one-shot kernel, world 1, %d floats: %.2f us per call" % (n, ev0.elapsed_time(ev1) * 1e3 / 500))
