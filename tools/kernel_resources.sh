#!/bin/bash
# Registers / scratch / occupancy of every kernel of rts_kernels.hip as the compiler reports them (cross-compile, no GPU).
cd "$(dirname "$0")/../raytracedshadows_amd/csrc" && make asm 2>&1 | python3 -c '
import re,sys
cur={}
for line in sys.stdin:
    m=re.search(r"(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if not m: continue
    k,v=m.group(1),m.group(2)
    if k in ("Function Name","Name"):
        cur={"name":v}
    cur[k]=v
    if k.startswith("LDS"):
        print("%-90s sgpr %3s vgpr %3s scratch %3s occ %s lds %s" % (cur["name"][:90], cur.get("TotalSGPRs"), cur.get("VGPRs"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), v))
'
