"""Workload for the SQ instruction / stall counters of the step's kernels (run under rocprofv3 --pmc ..., a few counters per pass):
the bench loop at the default configuration for HX_PMC_STEPS steps (default 200).  tools/pmc_summary.py turns the csv into per-kernel means."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402

loop = B.Loop(B.parse(sys.argv[1:]), 0, 1, torch.device("cuda", 0))
for _ in range(int(os.environ.get("HX_PMC_STEPS", 200))):
    loop.step()
torch.cuda.synchronize()
print("done")
