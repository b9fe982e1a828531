"""Per image of a rocprofv3 kernel trace of tools/tiled_batch.py: GPU span, busy union and time by kernel family.
python tools/tiled_trace_summary.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
first = [i for i, e in enumerate(ev) if 'tile_stats_kernel' in e[2] or 'tile_gather_kernel' in e[2]]
nms = [i for i, e in enumerate(ev) if 'nms_kernel' in e[2]]
# an image = 3 batches: group the batch-start markers in threes
for k in range(0, len(first) - 2, 3):
    i0 = first[k]
    nm = [i for i in nms if i > first[k + 2]][:1]
    if not nm:
        continue
    # last of the three nms launches of this image
    cand = [i for i in nms if i > i0][:3]
    i1 = cand[-1]
    seg = ev[i0:i1 + 1]
    start, end = seg[0][0], max(e[1] for e in seg)
    iv = sorted((s, e) for s, e, _ in seg)
    busy, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    fam = defaultdict(float)
    for s, e, n in seg:
        key = 'conv_bf16_pp' if 'conv_bf16_pp' in n else 'conv_bf16 (ring)' if 'conv_bf16_kernel' in n else 'conv_first' if 'conv_first' in n else \
              'nms' if 'nms' in n else 'tile gather / stats / zscore' if ('tile_' in n or 'zscore' in n) else 'transposes' if 'nchw' in n or 'nhwc' in n else \
              'decode' if 'decode' in n else 'torch elementwise / copies' if ('at::' in n or 'rocclr' in n) else 'other'
        fam[key] += (e - s) / 1e6
    print('image %d: GPU span %.2f ms, busy union %.2f ms, %d kernels; summed by family: %s' %
          (k // 3, (end - start) / 1e6, busy / 1e6, len(seg), ', '.join('%s %.2f' % kv for kv in sorted(fam.items(), key=lambda kv: -kv[1]))))
