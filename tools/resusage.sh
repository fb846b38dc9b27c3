#!/bin/bash
# usage: tools_resusage.sh file.hip  -> name vgpr sgpr scratch occupancy lds
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -c "$1" -o /tmp/_ru.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur={}
for line in sys.stdin:
    m=re.search(r'remark: (.*?) \[-Rpass', line)
    if not m: continue
    t=m.group(1)
    if t.startswith('Function Name:'):
        if cur: print(cur)
        cur={'fn':t.split(':',1)[1].strip()[:70]}
    else:
        k,v=t.split(':',1); 
        if k.strip() in ('VGPRs','AGPRs','TotalSGPRs','ScratchSize [bytes/lane]','Occupancy [waves/SIMD]','LDS Size [bytes/block]'): cur[k.strip().split(' ')[0]]=v.strip()
if cur: print(cur)
"
