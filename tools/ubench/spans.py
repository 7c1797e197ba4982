"""Where a step's time goes BETWEEN workgroups.  The phase-stamp build (`make -C hirl4ucav_amd/csrc stamps`) logs the life span of every
workgroup of every update / acting launch (s_memrealtime at its first instruction and at its exit, 10 ns ticks, one clock for the whole
device).  This tool runs a few steady-state steps, then logs one step with a critic-only learn() and one with an actor learn(), groups the
spans into launches (tag = kernel id, launches are serial on one stream) and prints per launch:

    first workgroup start | last workgroup start | first exit | last exit          (us, relative to the step's first workgroup)
    gap = this launch's first start - the previous launch's last exit              (drain + dispatch of the boundary)

    python3 tools/ubench/spans.py            (HX_STAMPS_LIB selects another stamps build)
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from hirl4ucav_amd import _lib  # noqa: E402

_lib.SO_PATH = os.environ.get("HX_STAMPS_LIB") or os.path.join(os.path.dirname(_lib.SO_PATH), "libhx_mi355_stamps.so")
import bench as B  # noqa: E402

CAP = 8192


def fetch(L):
    spans = np.zeros((CAP, 2), np.uint64)
    tags = np.zeros(CAP, np.uint32)
    n = ctypes.c_uint(0)
    rc = L.hx_debug_spans(spans.ctypes.data_as(ctypes.c_void_p), tags.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n))
    assert rc == 0, rc
    k = min(int(n.value), CAP)
    return spans[:k].astype(np.int64), tags[:k] & 0xFF  # (bits 8..: the workgroup's blockIdx — fetch_blocks)


def fetch_blocks(L):
    """as fetch(), with the workgroups' blockIdx.x / blockIdx.y"""
    spans = np.zeros((CAP, 2), np.uint64)
    tags = np.zeros(CAP, np.uint32)
    n = ctypes.c_uint(0)
    assert L.hx_debug_spans(spans.ctypes.data_as(ctypes.c_void_p), tags.ctypes.data_as(ctypes.c_void_p), ctypes.byref(n)) == 0
    k = min(int(n.value), CAP)
    return spans[:k].astype(np.int64), tags[:k] & 0xFF, (tags[:k] >> 8) & 0xFFFF, tags[:k] >> 24


def report(spans, tags, names):
    order = np.argsort(spans[:, 0], kind="stable")
    spans, tags = spans[order], tags[order]
    # a launch = a maximal run of spans whose starts lie before the earliest exit seen so far + slack, with tags of one kernel
    launches = []
    cur = [0]
    for i in range(1, len(spans)):
        same_kernel = names.get(int(tags[i]), ("?",))[0] == names.get(int(tags[cur[0]]), ("?",))[0]
        if same_kernel and spans[i, 0] < spans[cur, 1].max():
            cur.append(i)
        else:
            launches.append(cur)
            cur = [i]
    launches.append(cur)
    t0 = spans[0, 0]
    prev_end = None
    total_in, total_gap = 0.0, 0.0
    print("%-28s %5s | %8s %8s %8s %8s | %7s %7s | %6s" % ("launch", "wgs", "start0", "startN", "exit0", "exitN", "span", "wg max", "gap"))
    for idx in launches:
        s, e = spans[idx, 0], spans[idx, 1]
        name = names.get(int(tags[idx[0]]), ("line %d" % tags[idx[0]],))[0]
        gap = (s.min() - prev_end) / 100.0 if prev_end is not None else 0.0
        print("%-28s %5d | %8.2f %8.2f %8.2f %8.2f | %7.2f %7.2f | %6.2f" % (
            name, len(idx), (s.min() - t0) / 100.0, (s.max() - t0) / 100.0, (e.min() - t0) / 100.0, (e.max() - t0) / 100.0,
            (e.max() - s.min()) / 100.0, (e - s).max() / 100.0, gap))
        total_in += (e.max() - s.min()) / 100.0
        total_gap += gap
        prev_end = e.max()
    print("sum of launch spans %.2f us + gaps %.2f us = %.2f us (first workgroup start .. last workgroup exit)" % (
        total_in, total_gap, (spans[:, 1].max() - t0) / 100.0))


def main():
    loop = B.Loop(B.parse(["--no-front"]), 0, 1, torch.device("cuda", 0))  # the launches one by one (the front launch: tools/ubench/front_spans.py)
    L = _lib.load()
    L.hx_debug_spans.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint)]
    # tags are kernel ids (hx_update.h HX_SPAN_*)
    names = {1: ("fwd_l2",), 2: ("act_fused",), 3: ("bwd_l2",), 4: ("wgrad",)}
    for _ in range(40):
        loop.step()
    torch.cuda.synchronize()
    fetch(L)
    # five steps queued back to back (the host runs ahead of the GPU after the first ones), the last two reported: one with a critic-only
    # learn(), one with the delayed actor step
    for _ in range(5):
        loop.step()
    torch.cuda.synchronize()
    spans, tags = fetch(L)
    order = np.argsort(spans[:, 0], kind="stable")
    spans, tags = spans[order], tags[order]
    act_line = [ln for ln, lab in names.items() if lab[0] == "act_fused"][0]
    is_act = tags == act_line
    # first workgroup of each acting launch: an act span whose predecessor in time is not an act span
    starts = [i for i in range(len(spans)) if is_act[i] and (i == 0 or not is_act[i - 1])]
    starts.append(len(spans))
    for k in (len(starts) - 3, len(starts) - 2):
        a, b = starts[k], starts[k + 1]
        print("\n== step %d of 5 queued back to back (%d workgroups logged) ==" % (k, b - a))
        report(spans[a:b], tags[a:b], names)


if __name__ == "__main__":
    main()
