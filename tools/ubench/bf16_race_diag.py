"""diagnostic: which workspace fields differ between identical bf16 learn() calls on fresh engines"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from hirl4ucav_amd.agents import engine as E
from tests import _hirl_data as D
from tests.test_hirl_gpu import device_tables

XP, H1, H2, OW, KC = 20, 256, 512, 8, 8
per_row = XP + H1 + 2 + H1 + H2 + 2 + OW + H2 + H1 + OW + 2 * KC
names = ["TA", "C1", "C2", "TC1", "TC2", "API", "ABC", "BCS", "CPI", "CSOFT"]
fields = (("x", XP), ("z1", H1), ("st1", 2), ("h1", H1), ("z2", H2), ("st2", 2), ("outv", OW), ("dz2", H2), ("dh1", H1), ("dout", OW), ("lnp", 16))
B = 128
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
params, data = D.make_params(D.PARAM_SEED), D.make_data(D.DATA_SEED)
ring, exp, bc = device_tables(data)
rng = np.random.default_rng(0)
idx = rng.integers(0, D.N_REPLAY, B).astype(np.int32); ibc = rng.integers(0, D.N_EXPERT, B).astype(np.int32)
runs = []
for it in range(8):
    e = E.HirlEngine(batch=B)
    e.load_params(params["actor"], params["critic"], params["bc_actor"])
    e.set_update_dtype(mode)
    e.assemble(ring, torch.from_numpy(idx).cuda(), bc_table=bc, idx_bc=torch.from_numpy(ibc).cuda())
    e.learn(noise=torch.full((4,), 0.1).cuda(), bc_weight_now=100)
    torch.cuda.synchronize()
    runs.append((e.ws.cpu().numpy().copy(), e.losses_host()))
ref = runs[0][0]
for it in range(1, len(runs)):
    ws = runs[it][0]
    msgs = []
    for i, nm in enumerate(names):
        o = i * per_row * B
        for k, n in fields:
            a, b = ref[o:o + B * n].reshape(B, n), ws[o:o + B * n].reshape(B, n)
            if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
                bad = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
                msgs.append(f"{nm}.{k}: {len(bad)} words differ, rows {np.unique(bad[:,0])[:8]} cols {np.unique(bad[:,1])[:8]} maxabs {np.abs(a-b).max():.3e}")
            o += B * n
    print("run", it, "losses", runs[it][1][:2], "|", "; ".join(m.split(":")[0] + m.split("maxabs")[1] for m in msgs) if msgs else "identical workspace")
