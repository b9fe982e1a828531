"""HBM floor of the early rows of a bf16 layer table (tools/bf16_ab.py N 608 --layers): the 608^2 .. 152^2 stages, where a layer moves
more bytes than the matrix pipe needs time for.  Algorithmic bytes = input once + residual once (3x3 stride-1 layers of these stages
all close a residual block) + output once, bf16; the RGB layer reads fp32 x 4 channels.  floor = bytes / 8 TB/s.
usage: python tools/bf16_row_floors.py profiles/r04_bf16_layers_45x608.txt  > profiles/r04_bf16_row_floors.md"""
import re
import sys

rows, first_us, tiles = [], None, None
for line in open(sys.argv[1]):
    m = re.match(r'\s*(\d+)\s+(\d+)\s+(\d+)\s+(\d)\s+(\d)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)', line)
    if m:
        M, cin, cout, k, s, cnt = (int(m.group(i)) for i in range(1, 7))
        rows.append((M, cin, cout, k, s, cnt, float(m.group(8))))
    m = re.search(r"'y3_conv2d_first_bf16': ([\d.]+)", line)
    if m:
        first_us = float(m.group(1))
    m = re.search(r'bs(\d+) 608', line)
    if m:
        tiles = int(m.group(1))
print('| layer | launches | algorithmic MB | floor at 8 TB/s (us) | measured (us) | fraction of the floor rate |')
print('|---|---|---|---|---|---|')
if first_us and tiles:
    b = tiles * 608 * 608 * (16 + 64)
    print('| RGB 3->32 3x3, 608^2 (fp32 x 4 in, bf16 out) | 1 | %.0f | %.0f | %.1f | %.2f |' % (b / 1e6, b / 8e6, first_us, b / 8e6 / first_us))
for M, cin, cout, k, s, cnt, avg in sorted(rows, key=lambda r: -r[0] * (r[1] + r[2])):
    if M < 1000000:
        continue
    resid = k == 3 and s == 1
    b = (M * s * s * cin + M * cout * (2 if resid else 1)) * 2
    print('| %d->%d %dx%d s%d%s, M = %d | %d | %.0f | %.0f | %.1f | %.2f |' % (cin, cout, k, k, s, ' + residual' if resid else '', M, cnt, b / 1e6, b / 8e6, avg, b / 8e6 / avg))
