import os, sys, numpy as np, torch
sys.path.insert(0, ".")
from hirl4ucav_amd.agents import engine as E
from tests import _hirl_data as D
from tests.test_hirl_gpu import device_tables
params, data = D.make_params(21), D.make_data(22, outliers=True)
ring, exp, bc = device_tables(data)
rng = np.random.default_rng(5)
e = E.HirlEngine(batch=128)
e.load_params(params["actor"], params["critic"], params["bc_actor"])
out = {}
for k in range(int(sys.argv[2])):
    idx = rng.integers(0, D.N_REPLAY, 128).astype(np.int32)
    ibc = rng.integers(0, D.N_EXPERT, 128).astype(np.int32)
    noise = rng.normal(0, 0.2, 4).astype(np.float32)
    e.assemble(ring, torch.from_numpy(idx).cuda(), bc_table=bc, idx_bc=torch.from_numpy(ibc).cuda())
    e.learn(noise=torch.from_numpy(noise).cuda(), bc_weight_now=0.3, bc_warm_up_weight=0.05)
    torch.cuda.synchronize()
    for n in ("critic", "grad_critic", "m_critic", "actor", "grad_actor"):
        out[f"{n}_{k}"] = getattr(e, n).cpu().numpy().copy()
    out[f"ws_{k}"] = e.ws.cpu().numpy().copy()
np.savez(sys.argv[1], **out)
print("flags tail", e.ws[-64:].view(torch.int32).tolist()[:20], e.ws[-1:].view(torch.int32).tolist())
