"""How much of a vector step is host issue time?  Enqueue K steps without synchronising and compare the moment the host
is done issuing with the moment the GPU is done executing."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

for serial in (True, False):
    args = argparse.Namespace(envs=4096, batch=128, scenario="straight_line", agent="hirl", actions="policy", staged=False, overlap=not serial)
    loop = bench.Loop(args, 0, 1, torch.device("cuda", 0))
    for _ in range(300):
        loop.step()
    torch.cuda.synchronize()
    K = int(os.environ.get('K', '40'))
    t0 = time.perf_counter()
    for _ in range(K):
        loop.step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"serial={serial}: host issue {1e6 * (t1 - t0) / K:.1f} us/step, GPU done {1e6 * (t2 - t0) / K:.1f} us/step", flush=True)
