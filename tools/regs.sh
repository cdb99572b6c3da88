#!/bin/bash
# usage: tools/regs.sh file.hip  -> kernel, VGPRs, AGPRs, occupancy (waves/SIMD), LDS bytes
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -c "$1" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys, re, subprocess
cur = {}
rows = []
for ln in sys.stdin:
    m = re.search(r'Function Name: (\S+)', ln) or re.search(r' Name: (\S+)', ln)
    if m:
        cur = {'name': m.group(1)}; rows.append(cur); continue
    for key in ('VGPRs', 'AGPRs', 'Occupancy [waves/SIMD]', 'LDS Size [bytes/block]', 'ScratchSize [bytes/lane]', 'SGPRs'):
        m = re.search(re.escape(key) + r': (\d+)', ln)
        if m and cur is not None: cur[key] = int(m.group(1))
names = subprocess.run(['c++filt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = n.replace('capdec::', '').replace('(capdec::GemmParams)', '').replace('void ', '')
    print(f\"{n[:70]:70s} v{r.get('VGPRs',0):4d} a{r.get('AGPRs',0):3d} s{r.get('SGPRs',0):3d} occ{r.get('Occupancy [waves/SIMD]',0):2d} lds{r.get('LDS Size [bytes/block]',0):6d} scr{r.get('ScratchSize [bytes/lane]',0)}\")
"
