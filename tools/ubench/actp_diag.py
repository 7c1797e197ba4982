import os, sys
import numpy as np, torch
sys.path.insert(0, ".")
from hirl4ucav_amd.agents import engine as E
from tests import _hirl_data as D
dt = sys.argv[1]; n = int(sys.argv[2])
params = D.make_params(D.PARAM_SEED)
e = E.HirlEngine(batch=128); e.load_params(params["actor"], params["critic"], params["bc_actor"])
if dt != "f32": e.set_act_dtype(dt)
rng = np.random.default_rng(n)
obs = torch.from_numpy(rng.uniform(-1, 1, (n, 13)).astype(np.float32)).cuda()
per = torch.from_numpy(rng.normal(0, 0.3, (n, 4)).astype(np.float32)).cuda()
for name, kw in (("none", {}), ("per", dict(noise=per)), ("sigma", dict(sigma=0.1, seed=9, row0=77))):
    e.act_calls = 10
    a = e.act(obs, **kw).clone()
    os.environ["HX_ACT_PERSIST"] = "0"
    e.act_calls = 10
    b = e.act(obs, **kw).clone()
    os.environ.pop("HX_ACT_PERSIST")
    bad = (a.view(torch.int32) != b.view(torch.int32)).any(1).nonzero().flatten().cpu().numpy()
    print(dt, n, name, "mismatching rows:", len(bad), bad[:20], "max |d|", float((a - b).abs().max()))
    if len(bad):
        print("   rows mod 32:", sorted(set((bad % 32).tolist()))[:40], " tiles:", sorted(set((bad // 32).tolist()))[:10])
