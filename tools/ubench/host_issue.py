"""How much of a vector step is host issue time?  Enqueue K steps without synchronising and compare the moment the host
is done issuing with the moment the GPU is done executing — the front loop (bench.py's default) and the reference-order loop."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

for flags in ([], ["--no-front"]):
    loop = bench.Loop(bench.parse(flags), 0, 1, torch.device("cuda", 0))
    for _ in range(300):
        loop.step()
    torch.cuda.synchronize()
    for K in (20, 400):
        t0 = time.perf_counter()
        for _ in range(K):
            loop.step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{' '.join(flags) or 'front loop':12s} K={K:4d}: host issue {1e6 * (t1 - t0) / K:.1f} us/step, GPU done {1e6 * (t2 - t0) / K:.1f} us/step", flush=True)
