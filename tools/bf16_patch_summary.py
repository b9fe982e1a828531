"""HBM figures of the bf16 patch kernels (conv_bf16_c32_kernel / conv_bf16_c64_kernel) from three rocprofv3 passes over
`python3 tools/bf16_ab.py 45 608` (one planned batch of the tiled path): kernel trace (durations) + FETCH_SIZE + WRITE_SIZE.
usage: python tools/bf16_patch_summary.py <kernel_trace.csv> <fetch counter_collection.csv> <write counter_collection.csv>

ALGORITHMIC bytes per launch = input activations once + residual once + output once (bf16, 45 tiles of 608^2; the 36 / 147 KB of
weights are noise).  FETCH_SIZE x 2 (gfx950 reports half of wide coalesced reads: MI355X_MICROARCH.md, HBM section), both
counters KiB -> bytes.  frac = algorithmic bytes / average duration / 8 TB/s."""
import csv
import sys
from collections import defaultdict

MB = 1e6
T = 45
ROWS = {   # kernel-name fragment -> (label, algorithmic bytes)
    'conv_bf16_c32_kernel<2, false>': ('32->64 3x3 s2, 608^2 -> 304^2', T * 608 * 608 * 32 * 2 + T * 304 * 304 * 64 * 2),
    'conv_bf16_c32_kernel<1, true>': ('32->64 3x3 s1 + residual, 304^2', T * 304 * 304 * 32 * 2 + 2 * T * 304 * 304 * 64 * 2),
    'conv_bf16_c64_kernel<2>': ('64->128 3x3 s2, 304^2 -> 152^2', T * 304 * 304 * 64 * 2 + T * 152 * 152 * 128 * 2),
    'conv_bf16_c64_kernel<1>': ('64->128 3x3 s1 + residual, 152^2', T * 152 * 152 * 64 * 2 + 2 * T * 152 * 152 * 128 * 2),
}


def key_of(name):
    for k in ROWS:
        if k in name:
            return k
    return None


def load(path, counter=None, mult=1.0):
    acc = defaultdict(list)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = key_of(r['Kernel_Name'])
            if not k or (counter and r.get('Counter_Name') != counter):
                continue
            acc[k].append(float(r['Counter_Value']) * 1024.0 * mult if counter else float(int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    return acc


def main():
    trace, fetch, write = sys.argv[1:4]
    dur, fe, wr = load(trace), load(fetch, 'FETCH_SIZE', 2.0), load(write, 'WRITE_SIZE', 1.0)
    print('| kernel | layer | launches | avg us | algorithmic MB | TB/s (algorithmic / avg) | frac of 8 TB/s | FETCH x2 MB | WRITE MB | TB/s (counters / avg) |')
    print('|---|---|---|---|---|---|---|---|---|---|')
    for k, (label, ab) in ROWS.items():
        d = sorted(dur.get(k, []))
        if not d:
            continue
        d = d[:max(1, len(d) - 2)]
        avg = sum(d) / len(d) / 1e3
        f = sum(fe.get(k, [0])) / max(1, len(fe.get(k, [0])))
        w = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0])))
        print('| `%s` | %s | %d | %.1f | %.1f | %.2f | %.2f | %.1f | %.1f | %.2f |' % (k, label, len(d), avg, ab / MB, ab / avg / 1e6, ab / avg / 1e6 / 8.0, f / MB, w / MB, (f + w) / avg / 1e6))


if __name__ == '__main__':
    main()
