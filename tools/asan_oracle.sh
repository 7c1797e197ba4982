#!/bin/bash
# The CPU oracle under AddressSanitizer + UBSan: builds oracle/libhx_oracle_asan.so and runs the CPU tests that exercise the C restatement on
# it (golden vectors, opponent streams, batch stepping with the fused ring insert, the wire protocol's oracle backend).  -> profiles/<tag>_asan_oracle.log
set -uo pipefail
cd "$(dirname "$0")/.."
TAG=${1:-r04}
make -C oracle asan >/dev/null || exit 1
ASAN_RT=$(gcc -print-file-name=libasan.so)
LOG=profiles/${TAG}_asan_oracle.log
{
  echo "# $(date -u +%FT%TZ)  gcc $(gcc -dumpversion)  -fsanitize=address,undefined  (LD_PRELOAD=$ASAN_RT, detect_leaks=0: the interpreter's own allocations)"
  HX_ORACLE_LIBRARY=$PWD/oracle/libhx_oracle_asan.so LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
    python -m pytest tests/test_oracle_env.py tests/test_dynamics_sanity.py tests/test_sim_sample.py tests/test_wire_cpu.py -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -15
} | tee "$LOG"
grep -q "ERROR: AddressSanitizer\|runtime error" "$LOG" && { echo "SANITIZER FINDINGS"; exit 1; }
grep -q " passed" "$LOG"
