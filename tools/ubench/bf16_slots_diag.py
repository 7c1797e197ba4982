"""diagnostic: z2 / dh1 of every workspace slot after one bf16 learn() against a torch evaluation on rounded operands"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import torch.nn.functional as F
from hirl4ucav_amd.agents import engine as E
from tests import _hirl_data as D
from tests.test_hirl_gpu import device_tables

XP, H1, H2, OW, KC = 20, 256, 512, 8, 8
per_row = XP + H1 + 2 + H1 + H2 + 2 + OW + H2 + H1 + OW + 2 * KC
names = ["TA", "C1", "C2", "TC1", "TC2", "API", "ABC", "BCS", "CPI", "CSOFT"]
B = 128
params, data = D.make_params(D.PARAM_SEED), D.make_data(D.DATA_SEED)
ring, exp, bc = device_tables(data)
for mode in ("f32", "bf16"):
    e = E.HirlEngine(batch=B)
    e.load_params(params["actor"], params["critic"], params["bc_actor"])
    e.set_update_dtype(mode)
    rng = np.random.default_rng(0)
    idx = rng.integers(0, D.N_REPLAY, B).astype(np.int32); ibc = rng.integers(0, D.N_EXPERT, B).astype(np.int32)
    e.assemble(ring, torch.from_numpy(idx).cuda(), bc_table=bc, idx_bc=torch.from_numpy(ibc).cuda())
    noise = torch.zeros(4).cuda()
    e.learn(noise=noise, bc_weight_now=100)
    torch.cuda.synchronize()
    print(mode, "losses", e.losses_host(), "soft_count", int(e.soft_count.item()))
    for nz in (0.0, 0.3):
        e2 = E.HirlEngine(batch=B)
        e2.load_params(params["actor"], params["critic"], params["bc_actor"])
        e2.set_update_dtype(mode)
        e2.assemble(ring, torch.from_numpy(idx).cuda(), bc_table=bc, idx_bc=torch.from_numpy(ibc).cuda())
        e2.learn(noise=torch.full((4,), nz).cuda(), bc_weight_now=100)
        print(mode, "noise", nz, "losses", e2.losses_host(), "soft_count", int(e2.soft_count.item()))
    ws = e.ws.cpu().numpy()
    def slot(i):
        b = ws[i * per_row * B:(i + 1) * per_row * B]
        o = 0
        out = {}
        for k, n in (("x", XP), ("z1", H1), ("st1", 2), ("h1", H1), ("z2", H2), ("st2", 2), ("outv", OW), ("dz2", H2), ("dh1", H1), ("dout", OW), ("lnp", 16)):
            out[k] = b[o:o + B * n].reshape(B, n); o += B * n
        return out
    nets = {"TA": ("actor", None), "C1": ("critic", ("full1", "layernorm1", "full2")), "C2": ("critic", ("full3", "layernorm3", "full4")),
            "TC1": ("critic", ("full1", "layernorm1", "full2")), "TC2": ("critic", ("full3", "layernorm3", "full4")), "API": ("actor", None), "ABC": ("actor", None),
            "BCS": ("bc_actor", None)}
    for i, nm in enumerate(names[:8]):
        s = slot(i)
        which, keys = nets[nm]
        p = {k: torch.as_tensor(v) for k, v in params[which].items()}
        a, ln, b2 = keys if keys else ("full1", "layernorm1", "full2")
        x = torch.from_numpy(s["x"][:, :p[a + ".weight"].shape[1]])
        rows_np = data["replay"][idx]
        if nm == "TA":
            x = torch.from_numpy(rows_np[:, 17:30].copy())
        if nm == "BCS":
            x = torch.from_numpy(rows_np[:, 0:13].copy())
        if nm in ("TC1", "TC2"):
            from oracle import hirl_oracle as H
            import contextlib
            with (H.Bf16Layer2() if mode == "bf16" else contextlib.nullcontext()):
                na = H.actor_forward({k: torch.as_tensor(v) for k, v in params["actor"].items()}, torch.from_numpy(rows_np[:, 17:30].copy()))
            print(mode, "TA action max diff", float(np.abs(slot(0)["outv"][:, :4] - na.numpy()).max()))
            x = torch.cat([torch.from_numpy(rows_np[:, 17:30].copy()), na], 1)
        h = F.relu(F.layer_norm(F.linear(x, p[a + ".weight"], p[a + ".bias"]), (256,), p[ln + ".weight"], p[ln + ".bias"], 1e-5))
        w = p[b2 + ".weight"]
        if mode == "bf16":
            z2 = (h.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().t()).float() + p[b2 + ".bias"]
        else:
            z2 = F.linear(h, w, p[b2 + ".bias"])
        d = np.abs(s["z2"] - z2.numpy())
        print(mode, nm, "h1 max diff", float(np.abs(s["h1"] - h.numpy()).max()) if s["h1"].any() else "n/a (not saved)", "z2 max diff %.3e" % d.max(), "bad cols", np.unique(np.nonzero(d > 1e-2)[1])[:12], "bad rows", np.unique(np.nonzero(d > 1e-2)[0])[:12])
