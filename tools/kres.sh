#!/bin/bash
# kernel resource usage of one HIP translation unit: name, VGPRs, AGPRs, scratch, LDS
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=fast -fno-slp-vectorize -I../../include -Rpass-analysis=kernel-resource-usage -c $1 -o /tmp/kres_chk.o 2>&1 | python3 -c "
import sys,re
name=None
for l in sys.stdin:
    if 'error' in l: print(l.strip())
    m=re.search(r'Function Name: (\S+)',l)
    if m: name=m.group(1); vals={}
    for k in ('VGPRs','AGPRs','ScratchSize \[bytes/lane\]','LDS Size \[bytes/block\]','Occupancy \[waves/SIMD\]'):
        m=re.search(k+r': (\d+)',l)
        if m: vals[k.split(' ')[0]]=m.group(1)
    if 'LDS Size' in l and name:
        print(name[:60], vals)
"
