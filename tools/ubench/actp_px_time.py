"""Times the persistent acting launch for each phase-removal build made by tools/ubench/actp_variants.sh (HX_LIBRARY per child process)."""
import os
import subprocess
import sys

dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
for m in os.environ.get("MASKS", "0 1 2 4 8 32 12 35 47").split():
    env = dict(os.environ, HX_LIBRARY=os.path.abspath(f"hirl4ucav_amd/libhx_px{m}.so"), SIZES=os.environ.get("SIZES", "131072"))
    out = subprocess.run([sys.executable, "tools/ubench/actp_time.py", dt], env=env, capture_output=True, text=True).stdout
    print(f"HX_PX={m:>2s}", out.strip().splitlines()[-1] if out.strip() else "(no output)", flush=True)
