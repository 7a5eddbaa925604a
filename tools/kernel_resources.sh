#!/bin/bash
# Registers, scratch, LDS and occupancy of every kernel of the product library (compile-time report of hipcc):
#   tools/kernel_resources.sh [name filter]
here="$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I"$here/include" -c "$here/deepstructuredmixtures_amd/csrc/dsmgp_hip.cpp" -o /tmp/kres.o \
      -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re, sys
cur = None
rows = {}
for l in sys.stdin:
    m = re.search(r'remark: \s*(.*?) \[-Rpass', l)
    if not m: continue
    t = m.group(1)
    if t.startswith('Function Name:'):
        cur = t.split(':', 1)[1].strip(); rows[cur] = {}
    elif cur and ':' in t:
        k, v = t.split(':', 1); rows[cur][k.strip()] = v.strip()
import subprocess
flt = sys.argv[1] if len(sys.argv) > 1 else ''
for n, r in rows.items():
    d = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout.strip().split('(')[0]
    if flt in d:
        print(f\"{d:70s} VGPR {r.get('VGPRs','?'):>4} AGPR {r.get('AGPRs','?'):>3} SGPR {r.get('TotalSGPRs','?'):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>3} LDS {r.get('LDS Size [bytes/block]','?'):>6} occ {r.get('Occupancy [waves/SIMD]','?')}\")
" "$1"
