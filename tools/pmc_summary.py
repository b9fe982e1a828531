"""Aggregate a rocprofv3 counter_collection.csv by (kernel, grid): mean of every counter over the dispatches.
usage: python tools/pmc_summary.py <counter_collection.csv> [substring of kernel name]"""
import csv
import sys
from collections import defaultdict

path, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else '')
acc = defaultdict(lambda: defaultdict(list))
with open(path) as fh:
    for row in csv.DictReader(fh):
        name = row.get('Kernel_Name', '')
        if filt not in name:
            continue
        key = (name[:60], row.get('Grid_Size', ''), row.get('LDS_Block_Size', ''), row.get('VGPR_Count', ''), row.get('Accum_VGPR_Count', ''))
        acc[key][row['Counter_Name']].append(float(row['Counter_Value']))
for key, ctrs in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('GRBM_GUI_ACTIVE', [0]))):
    n = max(len(v) for v in ctrs.values())
    print('%s grid=%s lds=%s vgpr=%s agpr=%s dispatches=%d' % (*key, n))
    print('   ' + ' '.join('%s=%.4g' % (k, sum(v) / len(v)) for k, v in sorted(ctrs.items())))
