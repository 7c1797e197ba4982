"""Prints the distributions behind tests/test_bf16_update_gpu.py::test_bf16_free_running_against_the_fp32_reference_recording
(HX_PRINT_STATS=1 python -m pytest ... -s would do the same)."""
import os
import subprocess
import sys

os.environ["HX_PRINT_STATS"] = "1"
sys.exit(subprocess.call([sys.executable, "-m", "pytest", "tests/test_bf16_update_gpu.py", "-q", "-m", "gpu", "-s", "-k", "free_running"]))
