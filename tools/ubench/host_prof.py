import sys, time, collections
import torch
sys.path.insert(0, ".")
import bench
from hirl4ucav_amd import _lib
loop = bench.Loop(bench.parse(sys.argv[1:]), 0, 1, torch.device("cuda", 0))
for _ in range(300): loop.step()
torch.cuda.synchronize()
acc = collections.defaultdict(float); cnt = collections.Counter()
orig = _lib.call
def timed(name, *a):
    t = time.perf_counter(); r = orig(name, *a); acc[name] += time.perf_counter() - t; cnt[name] += 1; return r
_lib.call = timed
import hirl4ucav_amd.agents.engine as E
E._lib.call = timed
for K in (20, 20, 400):
    acc.clear(); cnt.clear()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K): loop.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"K={K}: host {1e6*(t1-t0)/K:.1f} us/step;", ", ".join(f"{k} {1e6*v/K:.1f}" for k, v in acc.items()), f"; python rest {1e6*((t1-t0)-sum(acc.values()))/K:.1f}")
