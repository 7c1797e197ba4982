#!/bin/bash
# VGPRs / scratch / LDS of every kernel in one .hip file (compiles it for gfx950 with -Rpass-analysis=kernel-resource-usage)
#   tools/kernel_resources.sh hirl4ucav_amd/csrc/hx_fwdbwd.hip [extra hipcc flags]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 "$@" -c "$f" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/ {n=$0; sub(/.*Function Name: /,"",n); sub(/ \[-Rpass.*/,"",n)}
       / VGPRs:/ {v=$0; sub(/.* VGPRs: /,"",v); sub(/ \[.*/,"",v)}
       /ScratchSize/ {s=$0; sub(/.*: /,"",s); sub(/ \[.*/,"",s)}
       /LDS Size/ {l=$0; sub(/.*: /,"",l); sub(/ \[.*/,"",l); printf "%-4s vgpr %-6s scratch %-8s lds  %s\n", v, s, l, n}' | c++filt | sed 's/(anonymous namespace):://g' | sort -k7
