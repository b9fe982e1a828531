"""Achieved HBM GB/s of the HBM-bound kernels of the path (decode, NMS, z-score, Adam, BatchNorm apply) from three
rocprofv3 passes over tools/hbm_path_driver.py: kernel trace (durations) + FETCH_SIZE + WRITE_SIZE counter passes.
usage: python tools/hbm_path_summary.py <kernel_trace.csv> <fetch counter_collection.csv> <write counter_collection.csv>  > profiles/rNN_hbm_path.md

Columns: launches; average duration (us); ALGORITHMIC bytes per launch (decode reads and writes N*Nb*(5+K)*4 B,
NMS reads the same rows, Adam 7 fp32 streams x 61.79 M, z-score 3 passes x 4 B per value) and the GB/s they give over the
average duration; FETCH_SIZE x 2 (gfx950 reports half of wide coalesced reads: MI355X_MICROARCH.md, HBM section) and
WRITE_SIZE per launch from the counters, both KiB -> bytes.  Small kernels move 1-3 MB: they are launch / latency bound and the
GB/s column says how far from the 8 TB/s (6.3 achievable) they necessarily sit; the us column is the figure to compare."""
import csv
import sys
from collections import defaultdict

KERNELS = ('decode_kernel', 'nms_kernel', 'zscore_partial_kernel', 'zscore_apply_kernel', 'adam_kernel', 'nchw_to_nhwc_kernel')
K, A = 2, 2
PARAMS = 61789770 + 288          # arena floats incl. the zero-padded RGB channel, alignment excluded


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


def algorithmic(kern, grid, wg):
    """bytes per launch, keyed by what the driver runs (batch 8 x 416^2 and 25 x 608^2)."""
    cfgs = {8: (416, 7098), 25: (608, 15162)}
    out = {}
    for n, (img, nb) in cfgs.items():
        rows = n * nb * (5 + K) * 4
        vals = n * 3 * img * img
        out[n] = {'decode_kernel': 2 * rows, 'nms_kernel': rows, 'zscore_partial_kernel': vals * 4, 'zscore_apply_kernel': vals * 8,
                  'nchw_to_nhwc_kernel': vals * 4 + n * img * img * 16, 'adam_kernel': 7 * PARAMS * 4}
    return out


def load(path, counter=None, mult=1.0):
    """rows of a rocprofv3 CSV in dispatch order -> {(kernel, grid, workgroup): [value, ...]} (durations in ns, or counter bytes)"""
    rows = []
    with open(path) as fh:
        for r in csv.DictReader(fh):
            k = short(r['Kernel_Name'])
            if not k or (counter and r.get('Counter_Name') != counter):
                continue
            # the trace reports per-axis sizes, the counter file their products
            grid = int(r['Grid_Size']) if 'Grid_Size' in r else int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
            wg = int(r['Workgroup_Size']) if 'Workgroup_Size' in r else int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']) * int(r['Workgroup_Size_Z'])
            val = float(r['Counter_Value']) * 1024.0 * mult if counter else float(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
            order = int(r['Start_Timestamp']) if not counter else int(r.get('Dispatch_Id', len(rows)))
            rows.append((order, (k, grid, wg), val))
    rows.sort()
    acc = defaultdict(list)
    for _, key, val in rows:
        acc[key].append(val)
    return acc


# the driver runs, per configuration, 12 launches of every kernel on the network's own rows, then 12 NMS launches on the sparse
# stress rows and 12 on the dense ones: split the NMS sequence of a grid into those thirds
NMS_CASES = ('rows of the random-init network (dense: every score near 0.5)', 'SURVEY 8d stress rows, sparse (obj = U^8)', 'SURVEY 8d stress rows, dense (obj = U)')


def main():
    trace, fetch, write = sys.argv[1:4]
    dur, fe, wr = load(trace), load(fetch, 'FETCH_SIZE', 2.0), load(write, 'WRITE_SIZE', 1.0)
    alg = algorithmic(None, None, None)
    by_kernel = defaultdict(list)
    for key in dur:
        by_kernel[key[0]].append(key)
    print('| kernel | config | launches | avg us | min us | algorithmic MB | GB/s (algorithmic bytes / avg) | FETCH x2 MB | WRITE MB | GB/s (counter bytes / avg) |')
    print('|---|---|---|---|---|---|---|---|---|---|')

    def line(k, cfg, d, ab, f, w):
        d = sorted(d)
        d = d[:max(1, len(d) - 2)]            # drop the two slowest (first-touch launches)
        avg = sum(d) / len(d) / 1e3
        fm = sum(f) / max(1, len(f))
        wm = sum(w) / max(1, len(w))
        print('| %s | %s | %d | %.1f | %.1f | %.3f | %.1f | %.3f | %.3f | %.1f |' % (k, cfg, len(d), avg, d[0] / 1e3, ab / 1e6, ab / avg / 1e3, fm / 1e6, wm / 1e6, (fm + wm) / avg / 1e3))

    for k in KERNELS:
        keys = sorted(by_kernel.get(k, []), key=lambda t: t[1])
        for i, key in enumerate(keys):
            n = 8 if (len(keys) == 1 or i < len(keys) / 2) else 25
            if k == 'adam_kernel':
                n = 8
            cfg = 'bs %d, grid %d x %d' % (n, key[1] // max(key[2], 1), key[2])
            if k == 'nms_kernel' and len(dur[key]) % 3 == 0:
                m = len(dur[key]) // 3
                for c in range(3):
                    sl = slice(c * m, (c + 1) * m)
                    line(k, cfg + ', ' + NMS_CASES[c], dur[key][sl], alg[n][k], fe.get(key, [0])[sl] or [0], wr.get(key, [0])[sl] or [0])
            else:
                line(k, cfg, dur[key], alg[n][k], fe.get(key, [0]), wr.get(key, [0]))


if __name__ == '__main__':
    main()
