"""Per-kernel view of the two PMC passes of tools/profile_round.sh / tools/traffic_pass.sh:
   python tools/traffic_by_kernel.py <fetch_counter_collection.csv> <write_counter_collection.csv>
FETCH_SIZE doubled (gfx950, MI355X_MICROARCH.md), KiB -> bytes; steps = nchw_to_nhwc_kernel launches of the fetch pass."""
import collections
import csv
import re
import sys


def agg(path, counter, mult):
    d = collections.defaultdict(lambda: [0.0, 0])
    passes = 0
    for row in csv.DictReader(open(path)):
        if row['Counter_Name'] != counter:
            continue
        n = row['Kernel_Name']
        if 'nchw_to_nhwc_kernel' in n:
            passes += 1
        n = re.sub(r'\(.*', '', n).replace('void ', '')[:64]
        d[n][0] += float(row['Counter_Value']) * 1024 * mult
        d[n][1] += 1
    return d, passes


f, p = agg(sys.argv[1], 'FETCH_SIZE', 2)
w, _ = agg(sys.argv[2], 'WRITE_SIZE', 1)
print('steps %d' % p)
print('%-64s %10s %9s %9s %12s %12s' % ('kernel', 'calls/step', 'read GB', 'write GB', 'read MB/call', 'write MB/call'))
for n in sorted(set(f) | set(w), key=lambda n: -(f[n][0] + w[n][0])):
    if (f[n][0] + w[n][0]) / p < 5e7:
        continue
    print('%-64s %10.1f %9.3f %9.3f %12.1f %12.1f' % (n, max(f[n][1], w[n][1]) / p, f[n][0] / p / 1e9, w[n][0] / p / 1e9,
                                                   f[n][0] / max(1, f[n][1]) / 1e6, w[n][0] / max(1, w[n][1]) / 1e6))
