#!/bin/bash
# dev: per-kernel register / spill / scratch table of one translation unit.  usage: tools/kernel_usage.sh csrc/file.hip [pattern] [-D...]
SRC=$1; PAT=${2:-.}; shift; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$SRC" -o /tmp/ku.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re
cur=None;rows={}
for l in sys.stdin:
    if 'error' in l: print(l.strip())
    m=re.search(r'remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|Occupancy \[waves/SIMD\]|SGPRs):\s+(\S+)',l)
    if not m: continue
    if m.group(1)=='Function Name': cur=m.group(2); rows[cur]={}
    elif cur: rows[cur][m.group(1)]=m.group(2)
import subprocess
for k,v in rows.items():
    if not re.search('$PAT',k): continue
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip().split('(')[0][:80]
    print('%-82s vgpr %3s agpr %3s spill %3s scratch %4s occ %s'%(name,v.get('VGPRs'),v.get('AGPRs'),v.get('VGPRs Spill'),v.get('ScratchSize [bytes/lane]'),v.get('Occupancy [waves/SIMD]')))
"
