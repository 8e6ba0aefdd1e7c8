#!/usr/bin/env python
"""Register / scratch usage of every kernel of one source: python scripts/kres.py csrc/conv_halo.hip [name filter]"""
import re, subprocess, sys, os
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ''
csrc = os.path.dirname(os.path.abspath(src))
out = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I', csrc, '-c', src, '-o', '/dev/null',
                      '-Rpass-analysis=kernel-resource-usage'] + sys.argv[3:], capture_output=True, text=True).stderr
cur = None
for l in out.splitlines():
    m = re.search(r'Function Name: (\S+)', l)
    if m:
        cur = {'name': m.group(1)}
        continue
    for key in ('VGPRs', 'AGPRs', 'VGPRs Spill', 'SGPRs Spill', 'ScratchSize \[bytes/lane\]', 'Occupancy \[waves/SIMD\]'):
        m = re.search(r'remark:\s+' + key + r': (\d+)', l)
        if m and cur is not None:
            cur[key.split(' [')[0].replace('\\', '')] = int(m.group(1))
    if cur and 'LDS Size' in l:
        if flt in cur['name']:
            n = subprocess.run(['c++filt', cur['name']], capture_output=True, text=True).stdout.strip().split('(')[0]
            print(f"{n:60s} VGPR {cur.get('VGPRs', 0):3d} AGPR {cur.get('AGPRs', 0):3d} spill {cur.get('VGPRs Spill', 0):3d} sgpr-spill {cur.get('SGPRs Spill', 0):3d} scratch {cur.get('ScratchSize', 0)}")
        cur = None
