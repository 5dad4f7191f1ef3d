#!/bin/bash
# Dev tool: per-kernel register / spill / LDS summary of one translation unit:  tools/kres.sh yk_exact.hip [extra flags]
cd "$(dirname "$0")/../k210_yolo_framework_amd/csrc" || exit 1
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. "$@" -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 |
  awk '/Function Name:/{n=$0; sub(/.*Function Name: /,"",n); sub(/ \[.*/,"",n)}
       /TotalSGPRs:/{s=$0; sub(/.*TotalSGPRs: /,"",s); sub(/ \[.*/,"",s)}
       / VGPRs:/{v=$0; sub(/.* VGPRs: /,"",v); sub(/ \[.*/,"",v)}
       /AGPRs:/{a=$0; sub(/.*AGPRs: /,"",a); sub(/ \[.*/,"",a)}
       /ScratchSize/{sc=$0; sub(/.*: /,"",sc); sub(/ \[.*/,"",sc)}
       /Occupancy/{o=$0; sub(/.*: /,"",o); sub(/ \[.*/,"",o)}
       /LDS Size/{l=$0; sub(/.*: /,"",l); sub(/ \[.*/,"",l); printf "%-90s sgpr %3s vgpr %3s agpr %3s scratch %4s occ %2s lds %s\n", n, s, v, a, sc, o, l}' | c++filt | sed 's/(anonymous namespace):://g'
