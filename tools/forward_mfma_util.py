"""Whole-forward MFMA utilisation from a rocprofv3 counter_collection.csv of tools/forward_mfma_driver.py (counters
SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE):  sum over the kernels of the forward of MFMA-busy cycles / (GPU-active cycles x 1024 SIMDs),
i.e. the time-weighted mean of the per-kernel MfmaUtil of tools/mfma_util.py.  python tools/forward_mfma_util.py <csv>"""
import csv
import sys
from collections import defaultdict

busy, act, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
for row in csv.DictReader(open(sys.argv[1])):
    name = row['Kernel_Name'].split('(')[0].replace('void ', '')
    fam = 'conv (MFMA)' if name.startswith('conv_') else 'library, other' if any(name.startswith(p) for p in ('bn_', 'decode', 'nchw', 'nhwc', 'upsample', 'slab', 'colsum')) else None
    if fam is None:
        continue            # torch's own fills / copies (model set-up)
    if row['Counter_Name'] == 'SQ_VALU_MFMA_BUSY_CYCLES':
        busy[fam] += float(row['Counter_Value'])
    elif row['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        act[fam] += float(row['Counter_Value']) / 8.0      # summed over the 8 XCDs
        cnt[fam] += 1
tb, ta = sum(busy.values()), sum(act.values())
print('| kernels of the fp32 forward (batch 8, 416 x 416) | launches | GPU-active cycles | MFMA utilisation |')
print('|---|---|---|---|')
for fam in act:
    print('| %s | %d | %.3g | %.1f %% |' % (fam, cnt[fam], act[fam], 100 * busy[fam] / (act[fam] * 1024)))
print('| **whole forward** | %d | %.3g | **%.1f %%** |' % (sum(cnt.values()), ta, 100 * tb / (ta * 1024)))
