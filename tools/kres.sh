#!/bin/bash
# usage: tools/kres.sh <file.hip> [grep-pattern]   -> kernel name, VGPRs, AGPRs, scratch, occupancy, LDS, spills
f=$1; pat=${2:-.}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -x hip -c "$f" -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur=None
def out(c):
    if c: print(c["n"][:120], "vgpr",c.get("VGPRs"), "agpr",c.get("AGPRs"), "scratch",c.get("ScratchSize [bytes/lane]"), "occ",c.get("Occupancy [waves/SIMD]"), "spill",c.get("VGPRs Spill"), "lds",c.get("LDS Size [bytes/block]"))
for l in sys.stdin:
    m=re.search(r"remark:\s+(.*?) \[-Rpass",l)
    if not m: continue
    t=m.group(1)
    k,v=t.split(":",1)
    if k=="Function Name":
        out(cur); cur={"n":v.strip()}
    elif cur is not None: cur[k.strip()]=v.strip()
out(cur)
' | grep -E "$pat"
rm -f /tmp/kres_$$.o
