"""Busy time of the conv kernel family from a rocprofv3 kernel trace: length of the UNION of the kernels' [start, end]
intervals (kernels of the two streams overlap, so their durations do not add up to wall time), per training step.
usage: python tools/trace_union.py <kernel_trace.csv> [steps executed]
Steps default to the number of nchw_to_nhwc_kernel launches in the trace: one per executed forward pass, i.e. warm-up +
timed steps + the instrumented passes of bench.py (round 1 passed 28 by hand where 27 had run)."""
import csv
import json
import sys

FAMILY = ('conv_igemm', 'conv_wgrad', 'conv_x3', 'splitk_epilogue', 'slab_reduce')    # splitk_epilogue: round 1 only; slab_reduce: kernel gradients with > 8 pixel splits
path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else None
spans, allspans, total, fwd_passes = [], [], 0, 0
with open(path) as fh:
    for row in csv.DictReader(fh):
        if 'nchw_to_nhwc_kernel' in row['Kernel_Name']:
            fwd_passes += 1
        s, e = int(row['Start_Timestamp']), int(row['End_Timestamp'])
        allspans.append((s, e))
        if any(f in row['Kernel_Name'] for f in FAMILY):
            spans.append((s, e))
            total += e - s


def union(sp):
    sp = sorted(sp)
    busy, cs, ce = 0, None, None
    for s, e in sp:
        if ce is None or s > ce:
            if ce is not None:
                busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if ce is not None:
        busy += ce - cs
    return busy


busy = union(spans)
busy_all = union(allspans)      # any kernel running: step time minus this is the GPU sitting idle between launches
if steps is None:
    steps = float(max(fwd_passes, 1))
print(json.dumps({'conv_family_kernels': len(spans), 'steps': steps, 'busy_union_ms_per_step': busy / steps * 1e-6,
                  'summed_durations_ms_per_step': total / steps * 1e-6, 'all_kernels_busy_union_ms_per_step': busy_all / steps * 1e-6}))
