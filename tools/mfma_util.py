"""MFMA utilisation per (kernel, grid) from a rocprofv3 counter_collection.csv holding SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE,
SQ_WAVE_CYCLES, SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY.
    MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)
GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS section), SQ_VALU_MFMA_BUSY_CYCLES over all SIMDs (checked:
it equals 64 cycles x the launch's MFMA count for v_mfma_f32_32x32x2_f32).  The SQ_WAIT_* / SQ_ACTIVE_* shares are of SQ_WAVE_CYCLES.
usage: python tools/mfma_util.py <counter_collection.csv> [substring of kernel name]"""
import csv
import sys
from collections import defaultdict

path, filt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'conv_')
acc = defaultdict(lambda: defaultdict(list))
with open(path) as fh:
    for row in csv.DictReader(fh):
        name = row.get('Kernel_Name', '')
        if filt not in name:
            continue
        key = (name.split('(')[0].replace('void ', ''), int(row.get('Grid_Size', 0)), int(row.get('Workgroup_Size', 256) or 256), row.get('VGPR_Count', ''))
        acc[key][row['Counter_Name']].append(float(row['Counter_Value']))
print('| kernel | workgroups | VGPRs | dispatches | MfmaUtil | parked (SQ_WAIT_ANY) | issue stall (SQ_WAIT_INST_ANY) | issuing (SQ_ACTIVE_INST_ANY) |')
print('|---|---|---|---|---|---|---|---|')
for key, c in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('SQ_VALU_MFMA_BUSY_CYCLES', [0]))):
    m = {k: sum(v) / len(v) for k, v in c.items()}
    if not m.get('GRBM_GUI_ACTIVE') or not m.get('SQ_WAVE_CYCLES'):
        continue
    util = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (m['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0)
    w = m['SQ_WAVE_CYCLES']
    print('| `%s` | %d | %s | %d | %.1f %% | %.0f %% | %.0f %% | %.0f %% |' % (
        key[0], key[1] // max(key[2], 1), key[3], len(c.get('GRBM_GUI_ACTIVE', [])), 100 * util, 100 * m.get('SQ_WAIT_ANY', 0) / w,
        100 * m.get('SQ_WAIT_INST_ANY', 0) / w, 100 * m.get('SQ_ACTIVE_INST_ANY', 0) / w))
