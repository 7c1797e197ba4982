import sys, torch
sys.path.insert(0, ".")
from hirl4ucav_amd import _lib
from hirl4ucav_amd.agents import sac_engine as SE
from hirl4ucav_amd.agents import engine as E
from tests.test_oracle_sac import sac_params
from tests import _hirl_data as D
p = sac_params()
e = SE.SacEngine(batch=128); e.load_params(p["policy"], p["q1"], p["q2"])
h = E.HirlEngine(batch=128); pp = D.make_params(1); h.load_params(pp["actor"], pp["critic"], pp["bc_actor"])
def t(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) * 1e3 / n
for n in (4096, 16384):
    obs = torch.rand(n, 13, device="cuda") * 2 - 1
    out = torch.empty(n, 4, device="cuda")
    st = _lib.stream_ptr()
    print(n, "SAC staged %.1f us" % t(lambda: _lib.call("hx_sac_act", e.policy.data_ptr(), obs.data_ptr(), n, out.data_ptr(), 2, None, 3, 0, 1, None, st)),
          "SAC image %.1f us" % t(lambda: _lib.call("hx_sac_act_f32i", e.policy.data_ptr(), e.w2_f32i.data_ptr(), obs.data_ptr(), n, out.data_ptr(), 2, None, 3, 0, 1, st)),
          "HIRL staged %.1f us" % t(lambda: _lib.call("hx_actor_act", h.actor.data_ptr(), obs.data_ptr(), n, out.data_ptr(), 3, None, 0.1, 1, 0, 1, 0.0, None, st)),
          "HIRL image %.1f us" % t(lambda: _lib.call("hx_actor_act_f32i", h.actor.data_ptr(), h.w2_f32i.data_ptr(), obs.data_ptr(), n, out.data_ptr(), 3, None, 0.1, 1, 0, 1, 0.0, st)))
n = 4096
obs = torch.rand(n, 13, device="cuda") * 2 - 1
out = torch.empty(n, 4, device="cuda")
st = _lib.stream_ptr()
for mode in (0, 2):
    print("SAC n=4096 mode", mode, "staged %.1f us" % t(lambda: _lib.call("hx_sac_act", e.policy.data_ptr(), obs.data_ptr(), n, out.data_ptr(), mode, None, 3, 0, 1, None, st)),
          "image %.1f us" % t(lambda: _lib.call("hx_sac_act_f32i", e.policy.data_ptr(), e.w2_f32i.data_ptr(), obs.data_ptr(), n, out.data_ptr(), mode, None, 3, 0, 1, st)))
for mode, sig in ((0, 0.0), (3, 0.1)):
    print("HIRL n=4096 noise mode", mode, "staged %.1f us" % t(lambda: _lib.call("hx_actor_act", h.actor.data_ptr(), obs.data_ptr(), n, out.data_ptr(), mode, None, sig, 1, 0, 1, 0.0, None, st)),
          "image %.1f us" % t(lambda: _lib.call("hx_actor_act_f32i", h.actor.data_ptr(), h.w2_f32i.data_ptr(), obs.data_ptr(), n, out.data_ptr(), mode, None, sig, 1, 0, 1, 0.0, st)))
