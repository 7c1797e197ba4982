"""Inside the FRONT launch (hx_front.hip): when do the acting workgroups, launch A's and launch B's workgroups start and end?  The phase-stamp build
(`make -C hirl4ucav_amd/csrc stamps`) logs every workgroup's first instruction and exit (s_memrealtime, 10 ns ticks); this tool queues a few
steps of `bench.py --front`, takes the last two (one critic-only learn(), one with the delayed actor step) and prints, per role and per launch behind
it, the workgroup count and first / last start and first / last exit in us from the step's first workgroup, plus a 2-us histogram of the starts.
    python3 tools/ubench/front_spans.py [bench.py flags, e.g. --envs 4096]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
sys.path.insert(0, "tools/ubench")
from hirl4ucav_amd import _lib  # noqa: E402

_lib.SO_PATH = os.environ.get("HX_STAMPS_LIB") or os.path.join(os.path.dirname(_lib.SO_PATH), "libhx_mi355_stamps.so")
import bench as B  # noqa: E402
from spans import fetch  # noqa: E402

NAMES = {1: "fwd_l2 (own launch)", 2: "front: act + env", 3: "bwd_l2", 4: "wgrad", 5: "front: launch A", 6: "front: launch B"}


def main():
    loop = B.Loop(B.parse(["--front"] + sys.argv[1:]), 0, 1, torch.device("cuda", 0))
    L = _lib.load()
    L.hx_debug_spans.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint)]
    for _ in range(40):
        loop.step()
    torch.cuda.synchronize()
    fetch(L)
    for _ in range(5):
        loop.step()
    torch.cuda.synchronize()
    spans, tags = fetch(L)
    order = np.argsort(spans[:, 0], kind="stable")
    spans, tags = spans[order], tags[order]
    act = np.flatnonzero(tags == 2)
    starts = [act[0]]
    for i in act[1:]:
        if spans[i, 0] - spans[starts[-1], 0] > 3000:  # > 30 us after the previous step's first acting workgroup: the next step
            starts.append(i)
    starts.append(len(spans))
    for k in (len(starts) - 3, len(starts) - 2):
        a, b = starts[k], starts[k + 1]
        s, t = spans[a:b], tags[a:b]
        t0 = s[:, 0].min()
        print("\n== step %d of 5 queued back to back: %d workgroups, %.2f us from the first start to the last exit ==" % (k, b - a, (s[:, 1].max() - t0) / 100.0))
        print("%-22s %5s | %8s %8s %8s %8s | %8s | starts per 2 us" % ("role / launch", "wgs", "start0", "startN", "exit0", "exitN", "wg mean"))
        for tag in (2, 5, 6, 3, 4, 1):
            m = t == tag
            if not m.any():
                continue
            st, en = (s[m, 0] - t0) / 100.0, (s[m, 1] - t0) / 100.0
            hist = np.bincount((st // 2).astype(int))
            print("%-22s %5d | %8.2f %8.2f %8.2f %8.2f | %8.2f | %s" % (NAMES[tag], m.sum(), st.min(), st.max(), en.min(), en.max(), (en - st).mean(),
                                                                      " ".join("%d:%d" % (2 * i, c) for i, c in enumerate(hist) if c)))
    stamps(loop, L)


def stamps(loop, L):
    """phase stamps (x 10 ns) of one workgroup of each role inside the front launch, critic-only and actor step"""
    out = np.zeros(80, np.float32)
    for k in range(4):
        loop.step()
        torch.cuda.synchronize()
        assert L.hx_debug_stamps_front(out.ctypes.data_as(ctypes.c_void_p)) == 0
        print("front launch, step kind %s (x10ns)" % ("actor" if not loop.eng.actor_trainable else "critic-only"))
        print("  launch A wg: draw + all requests issued %d | (head) %d | staging + sync (loads waited) %d | z1 %d | stats %d | norm %d | mfma+store %d" % tuple(out[9:16].tolist()))
        print("  launch B wg: requests issued + wait + x tile %d | previous net's head %d | staging + sync %d | z1 %d | stats %d | norm %d | mfma+store %d" % tuple(out[1:8].tolist()))
        print("  acting wg:   prologue %d | mfma %d | head+sync %d | env step %d | reset/store/obs %d | stats %d | sync %d | ring+obs out %d" % tuple(out[57:65].tolist()))


if __name__ == "__main__":
    main()
