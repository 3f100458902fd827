#!/bin/bash
# Register / LDS / occupancy figures of every kernel of one HIP source, as the compiler reports them
# (-Rpass-analysis=kernel-resource-usage).  usage: tools/kernel_resources.sh gingr_amd/csrc/affinity.hip [name filter]
SRC=$1; PAT=${2:-.}
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -c $SRC -o /dev/null \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
cur=None; rows={}
for line in sys.stdin:
    m=re.search(r'remark: .*Function Name: (\S+)',line)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r'remark:\s+([A-Za-z ]+\w)\s*(?:\[[^\]]*\])?: (\S+)',line)
    if m and cur: rows[cur][m.group(1).strip()]=m.group(2)
import subprocess
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()
    if re.search(r'$PAT',name):
        print(f\"{name[:80]:80s} VGPR={v.get('VGPRs','?'):>4s} AGPR={v.get('AGPRs','?'):>3s} SGPR={v.get('TotalSGPRs','?'):>4s} spill(v/s)={v.get('VGPRs Spill','?')}/{v.get('SGPRs Spill','?')} scratch={v.get('ScratchSize','?')} occ={v.get('Occupancy','?')} LDS={v.get('LDS Size','?')}\")
"
