#!/bin/bash
# usage: tools/kres.sh <file.hip> [extra hipcc flags]  -> kernel name, VGPRs, SGPRs, scratch, LDS, occupancy
f=$1; shift
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off "$@" -Rpass-analysis=kernel-resource-usage -c "$f" -o /tmp/kres.o 2>&1 | python3 -c "
import sys,re
cur=None
for line in sys.stdin:
    if 'error' in line: print(line.strip())
    m=re.search(r'Function Name: (\S+)',line)
    if m: cur={'name':m.group(1)}; continue
    if cur is None: continue
    for key,pat in (('vgpr',r' VGPRs: (\d+)'),('agpr',r'AGPRs: (\d+)'),('sgpr',r'TotalSGPRs: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)'),('lds',r'LDS Size \[bytes/block\]: (\d+)')):
        m=re.search(pat,line)
        if m: cur[key]=m.group(1)
    if 'lds' in cur:
        import subprocess
        name=subprocess.run(['c++filt',cur['name']],capture_output=True,text=True).stdout.strip()[:70]
        print(f\"{name:70s} vgpr={cur.get('vgpr'):>4s} agpr={cur.get('agpr','0'):>3s} sgpr={cur.get('sgpr'):>4s} scratch={cur.get('scratch'):>4s} lds={cur.get('lds'):>6s} occ={cur.get('occ')}\")
        cur=None
"
