#!/bin/bash
# per-kernel register / LDS / occupancy table of one HIP source (compiler view): tests/kres.sh lisreg_assoc.hip [extra flags]
F=${1:-lisreg_assoc.hip}; shift
cd "$(dirname "$0")/../lis-slam_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage "$@" -c $F -o /tmp/kres.o 2>&1 | python3 -c '
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r"Function Name: (\S+)",l)
    if m: cur=m.group(1); print(); print(cur[:110],end=" | ")
    for k in ("VGPRs:","VGPRs Spill","TotalSGPRs","Occupancy","LDS Size","ScratchSize"):
        m=re.search(re.escape(k)+r"[^:]*: (\d+)",l) if k!="VGPRs:" else re.search(r" VGPRs: (\d+)",l)
        if m: print(k.strip(":"),m.group(1),end=" | ")
print()'
