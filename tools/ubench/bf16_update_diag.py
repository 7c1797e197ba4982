import sys, os
sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
import torch  # noqa: F401 (CUDA context for the imports below)
from tests import test_bf16_update_gpu as T
from hirl4ucav_amd.agents import engine as E
gd = os.path.join("tests", "golden")
for mode in T.MODES:
    col = []
    T.run_mode(E, mode, gd, collect=col)
    print(col[-1])
    for what, got, ref, gc, ga in col[:-1]:
        lrel = max(abs(a - b) / max(abs(b), 1e-3) for a, b in zip(got, ref))
        wc = max(gc, key=lambda t: t[2]); fc = max(gc, key=lambda t: t[1])
        s = f"{what}: loss rel {lrel:.2e} | critic worst-loose {wc[0]} {wc[2]:.3f}, worst tight-frac {fc[0]} {fc[1]:.4f}"
        if ga:
            wa = max(ga, key=lambda t: t[2]); fa = max(ga, key=lambda t: t[1])
            s += f" | actor worst-loose {wa[0]} {wa[2]:.3f}, tight-frac {fa[0]} {fa[1]:.4f}"
        print(s, flush=True)
