"""Phase stamps of one fwd_l2 / bwd_l2 / wgrad / act_fused workgroup.  Diagnostic build (never the product .so):
    make -C hirl4ucav_amd/csrc stamps      (-> hirl4ucav_amd/libhx_mi355_stamps.so)
The layer-1 fields of the wgrad line are filled only by a build with -DHX_STAMPS_L1 (a dozen stamps inside that path double its length: the default stamps
build leaves them out so that workgroup life spans stay comparable, tools/ubench/wgrad_blocks.py).
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from hirl4ucav_amd import _lib  # noqa: E402

_lib.SO_PATH = os.path.join(os.path.dirname(_lib.SO_PATH), "libhx_mi355_stamps.so")
import bench as B  # noqa: E402
from hirl4ucav_amd.agents.engine import HxBatch  # noqa: E402

loop = B.Loop(B.parse(["--no-front"]), 0, 1, torch.device("cuda", 0))  # the launches one by one (the front launch: tools/ubench/front_spans.py)
for _ in range(20):
    loop.step()
e = loop.eng
e.sample(loop.replay, loop.expert, loop.bc_table, n_main=128, seed=2)
batch = HxBatch(e.rows.data_ptr(), e.bc_rows.data_ptr(), 128, e._noise.data_ptr())
L = _lib.load()
out = np.zeros(80, np.float32)
for i in range(3):
    # the one-call path: wgrad carries the optimizer step (critic-only call: critic_step i + 1, no actor phase)
    _lib.call("hx_hirl_learn", ctypes.byref(e.nets), ctypes.byref(batch), ctypes.byref(e.hyper), e.critic_step + i + 1, 0, 0, 0, 0, 0.0, 0.0, _lib.stream_ptr())
    torch.cuda.synchronize()
    assert L.hx_debug_stamps(out.ctypes.data_as(ctypes.c_void_p)) == 0
    print("wgrad<ADAM>(critic) x10ns: tiles: w known %d | loads+mfma %d store %d | vector: w known %d | loop %d reduce+store %d | layer1: w known %d | path entered %d | z1/dh1 issued %d | x issued %d | st1/lnp issued %d | adam issued %d | first barrier %d | staged (loads waited) %d | compute %d | partials to LDS %d | barrier %d | sum %d | Adam + stores %d" % (out[33], out[34], out[35], out[41], out[42], out[43], out[49], out[50], out[51], out[52], out[53], out[54], out[55], out[56], out[57], out[58], out[59], out[60], out[61]))
    print("fwd A, with the draw (x10ns): draw + all requests issued %d | (previous net's head) %d | staging + sync (loads waited) %d | z1 %d | stats %d | norm %d | mfma+store %d" % tuple(out[9:16].tolist()))
    print("fwd B (x10ns): requests issued %d | previous net's head %d | staging + sync (loads waited) %d | z1 %d | stats %d | norm %d | mfma+store %d      bwd: issue+wait %d | prologue %d | lossred %d | mfma %d | epilogue %d" % tuple(out[1:8].tolist() + out[17:22].tolist()))
for dt in ("f32", "bf16"):
    e.set_act_dtype(dt)
    for i in range(3):
        e.act(loop.env.obs, sigma=0.1, seed=1, out=loop.actions)
        torch.cuda.synchronize()
        assert L.hx_debug_stamps(out.ctypes.data_as(ctypes.c_void_p)) == 0
        print(dt, "act_fused x10ns (one workgroup of 256): prologue %d | mfma %d | head %d" % tuple(out[57:60].tolist()))
    for i in range(2):
        e.act_step(loop.env, sigma=0.1, seed=1, out=loop.actions)
        torch.cuda.synchronize()
        assert L.hx_debug_stamps(out.ctypes.data_as(ctypes.c_void_p)) == 0
        print(dt, "act+env x10ns: prologue %d | mfma %d | head+sync %d | env step %d | reset/store/obs %d | stats %d | sync %d | ring+obs out %d" % tuple(out[57:65].tolist()))
