"""Phase stamps of one fwd_l2 and one bwd_l2 workgroup (diagnostic build: make -C hirl4ucav_amd/csrc clean all STAMPS=1)."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, ".")
import bench as B
from hirl4ucav_amd import _lib
from hirl4ucav_amd.agents.engine import HxBatch
class A: envs=4096; batch=128; scenario="straight_line"
loop = B.Loop(A, 0, 1, torch.device("cuda", 0))
for _ in range(20): loop.step()
e = loop.eng
e.sample(loop.replay, loop.expert, loop.bc_table, n_main=128, seed=2)
batch = HxBatch(e.rows.data_ptr(), e.bc_rows.data_ptr(), 128, e._noise.data_ptr())
L = _lib.load(); out = np.zeros(64, np.float32)
for i in range(3):
    _lib.call("hx_hirl_critic_grads", ctypes.byref(e.nets), ctypes.byref(batch), ctypes.byref(e.hyper), 0, _lib.stream_ptr())
    torch.cuda.synchronize()
    assert L.hx_debug_stamps(out.ctypes.data_as(ctypes.c_void_p)) == 0
    print("wgrad(critic) x10ns: tiles: loads+mfma %d store %d | vector: loop %d reduce+store %d | layer1: loop %d reduce+store %d" % (out[33], out[34], out[41], out[42], out[49], out[50]))
    print("fwd (x10ns): W1 issue %d | zero+sync %d | gather+sync %d | z1 %d | stats %d | norm %d | mfma+store %d      bwd: issue+wait %d | prologue %d | lossred %d | mfma %d | epilogue %d" % tuple(out[1:8].tolist() + out[17:22].tolist()))
for i in range(3):
    e.act(loop.env.obs, sigma=0.1, seed=1, out=loop.actions)
    torch.cuda.synchronize()
    assert L.hx_debug_stamps(out.ctypes.data_as(ctypes.c_void_p)) == 0
    print("act_fused x10ns (one workgroup of 256): prologue %d | mfma %d | head %d" % tuple(out[57:60].tolist()))
