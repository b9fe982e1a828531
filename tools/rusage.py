"""Kernel resource table from a `hipcc -Rpass-analysis=kernel-resource-usage` log:
   python tools/rusage.py build.log   ->   name  sgpr vgpr agpr scratch occupancy lds"""
import re
import subprocess
import sys

KEYS = r'(Function Name|Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\])'
rows, cur = [], None
for line in open(sys.argv[1], errors='replace'):
    m = re.search(r': +' + KEYS + r': +(\S+)', line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k in ('Function Name', 'Name'):
        cur = {'name': v}
        rows.append(cur)
    elif cur is not None:
        cur[k.split(' ')[0]] = v
names = subprocess.run(['c++filt'] + [r['name'] for r in rows], capture_output=True, text=True).stdout.split('\n') if rows else []
for r, name in zip(rows, names):
    name = re.sub(r'\((FastArgs4?|ConvArgs|WgradArgs|[^)]*)\)$', '', name).replace('void ', '')
    print('%-84s s%-4s v%-4s a%-4s scr%-4s occ%-2s lds%s' % (name[:84], r.get('TotalSGPRs', '?'), r.get('VGPRs', '?'), r.get('AGPRs', '?'),
                                                           r.get('ScratchSize', '?'), r.get('Occupancy', '?'), r.get('LDS', '?')))
