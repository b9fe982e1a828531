"""Busy time of the conv kernel family from a rocprofv3 kernel trace: length of the UNION of the kernels' [start, end]
intervals (kernels of the two streams overlap, so their durations do not add up to wall time), per training step.
usage: python tools/trace_union.py <kernel_trace.csv> <steps executed>"""
import csv
import json
import sys

FAMILY = ('conv_igemm', 'conv_wgrad', 'splitk_epilogue', 'slab_reduce')
path, steps = sys.argv[1], float(sys.argv[2])
spans, total = [], 0
with open(path) as fh:
    for row in csv.DictReader(fh):
        if any(f in row['Kernel_Name'] for f in FAMILY):
            s, e = int(row['Start_Timestamp']), int(row['End_Timestamp'])
            spans.append((s, e))
            total += e - s
spans.sort()
busy, cs, ce = 0, None, None
for s, e in spans:
    if ce is None or s > ce:
        if ce is not None:
            busy += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
if ce is not None:
    busy += ce - cs
print(json.dumps({'conv_family_kernels': len(spans), 'steps': steps, 'busy_union_ms_per_step': busy / steps * 1e-6,
                  'summed_durations_ms_per_step': total / steps * 1e-6}))
