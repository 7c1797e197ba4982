#!/bin/bash
# The HOST side of libhx_mi355.so under AddressSanitizer on the CPU box (VERDICT r5 "a sanitizer target for the product's host code"):
# builds hirl4ucav_amd/libhx_mi355_asanhost.so (make asan-host: --offload-host-only, -fsanitize=address, launches are dry runs) and drives every entry
# point's host path with tools/asan_host_drive.py.  -> profiles/<tag>_asan_host.log.  CPU only: the GPU pool refuses sanitizer runs.
set -uo pipefail
cd "$(dirname "$0")/.."
TAG=${1:-r06}
make -C hirl4ucav_amd/csrc -j8 asan-host >/dev/null || exit 1
ASAN_RT=$(find /opt/rocm/lib/llvm/lib/clang -name 'libclang_rt.asan-x86_64.so' | head -1)
LOG=profiles/${TAG}_asan_host.log
{
  echo "# $(date -u +%FT%TZ)  hipcc --offload-host-only -DHX_HOST_DRYRUN -fsanitize=address  (LD_PRELOAD=$ASAN_RT, detect_leaks=0: the interpreter's own allocations)"
  HX_LIBRARY=$PWD/hirl4ucav_amd/libhx_mi355_asanhost.so LD_PRELOAD=$ASAN_RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1:alloc_dealloc_mismatch=0 \
    python tools/asan_host_drive.py 2>&1 | grep -v "amdgpu.ids" | tail -25
} | tee "$LOG"
grep -q "ERROR: AddressSanitizer" "$LOG" && { echo "SANITIZER FINDINGS"; exit 1; }
grep -q "ran their host paths" "$LOG"
