#!/bin/bash
# Which phase of the persistent acting kernel is on the critical path: builds of hx_actp.hip with one phase removed each (-DHX_PX=mask, wrong
# results, timing only), timed on one box against the full kernel.   tools/ubench/actp_variants.sh [dtype] [sizes]
set -e
cd "$(dirname "$0")/../.."
C=hirl4ucav_amd/csrc
OBJS=$(cd $C && ls hx_*.o | grep -v '^stamps_' | grep -v hx_actp.o | sed "s#^#$C/#")
for m in ${MASKS:-0 1 2 4 8 32 12 35 47}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DHX_PX=$m -c $C/hx_actp.hip -o /tmp/actp_px$m.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o hirl4ucav_amd/libhx_px$m.so $OBJS /tmp/actp_px$m.o
done
