"""Would hipGraph replay shorten the GPU-side chain?  Capture one actor-phase and one critic-only vector step (act -> env step ->
sample -> learn) with FROZEN call counters (wrong numerics on replay: counters key the Philox streams and Adam's bias correction —
this is a timing probe only) and compare replaying them alternately with issuing the same launches on the stream."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

args = argparse.Namespace(envs=4096, batch=128, scenario="straight_line", agent="hirl", actions="policy", staged=False, overlap=False,
                          type="soft", bc_weight=0.5)
loop = bench.Loop(args, 0, 1, torch.device("cuda", 0))
for _ in range(300):
    loop.step()
torch.cuda.synchronize()
K = 1000
t0 = time.perf_counter()
for _ in range(K):
    loop.step()
torch.cuda.synchronize()
print(f"stream launches: {1e6 * (time.perf_counter() - t0) / K:.1f} us/step", flush=True)

graphs = []
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for phase in range(2):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            loop.step()
        graphs.append(g)
torch.cuda.synchronize()
for _ in range(50):
    for g in graphs:
        g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K // 2):
    for g in graphs:
        g.replay()
torch.cuda.synchronize()
print(f"graph replay:    {1e6 * (time.perf_counter() - t0) / K:.1f} us/step", flush=True)
