"""Upper bound of a two-stream schedule: learn() (reference-order launches, one C call) back to back on one stream, act + env back to back on another,
NO dependencies between them (a throughput experiment only: the two race on the ring).  If the learn stream beside the acting stream does not run at
about its stand-alone pace, acting one step ahead on a second stream cannot beat the front launch (profiles/r05_two_stream_free_run.txt).
    python tools/ubench/two_stream_free_run.py [envs]"""
import sys
import time

import torch

sys.path.insert(0, ".")
import bench as B  # noqa: E402

n = sys.argv[1] if len(sys.argv) > 1 else "4096"
loop = B.Loop(B.parse(["--no-front", "--envs", n]), 0, 1, torch.device("cuda", 0))
for _ in range(200):
    loop.step()
torch.cuda.synchronize()
e = loop.eng
K = 4000


def learn_only():
    e.sample(loop.replay, loop.expert, loop.bc_table, n_main=e.batch - loop.expert_num, seed=2, defer=True)
    e.learn(bc_weight_now=None, bc_warm_up_weight=0.0)


def run(fn, k=K):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(k):
        fn()
    h = time.perf_counter() - t
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / k * 1e6, h / k * 1e6


for rep in range(3):
    a = run(loop._act_env)
    l = run(learn_only)
    s = run(loop.step)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def both():
        with torch.cuda.stream(s1):
            loop._act_env()
        with torch.cuda.stream(s2):
            learn_only()
    c = run(both)
    print("act+env alone %.2f us (host issue %.2f) | learn alone %.2f (%.2f) | serial step %.2f (%.2f) | two free streams %.2f per pair (%.2f)" % (a + l + s + c), flush=True)
