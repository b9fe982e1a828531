"""Per conv shape of layer_times.txt: the HBM floor (compulsory bytes at 4.5 TB/s), the MFMA floor (157.3 TFLOP/s nominal) and the
TFLOP/s each floor allows, beside the measured figure.  python tools/row_ceilings.py profiles/r03_layer_times.txt [min_tflops]
Rows below min_tflops (default 70) are listed: the table answers "why is this row under 70 TFLOP/s"."""
import sys

HBM = 4.5e12       # sustained read + write rate of the streaming kernels of this repo (profiles/r03_hbm_path.md: adam 5.06 TB/s)
PEAK = 157.3e12
rows = []
for line in open(sys.argv[1]):
    f = line.split()
    if len(f) == 10 and f[0] in ('fwd', 'dgrad', 'dgradb', 'wgrad'):
        kind, m, cin, cout, k, s = f[0], int(f[1]), int(f[2]), int(f[3]), int(f[4]), int(f[5])
        cnt, avg_us, tf = int(f[6]), float(f[8]), float(f[9])
        m_in = m * s * s                      # M counts OUTPUT pixels of the forward layer
        x, y = m_in * cin * 4, m * cout * 4
        flops = 2.0 * m * cin * cout * k * k
        if kind == 'fwd':
            byts = x + y
        elif kind == 'wgrad':
            byts = x + y
        elif kind == 'dgrad':
            byts = y + x
        else:                                 # data gradient + BatchNorm-backward moments of the layer below: reads its activation too
            byts = y + 2 * x
        t_h, t_m = byts / HBM * 1e6, flops / PEAK * 1e6
        rows.append((tf, kind, m, cin, cout, k, s, cnt, avg_us, t_h, t_m, flops / max(t_h, t_m) / 1e6))
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 70.0
print('| kind | M | Cin→Cout | k/s | launches | measured µs | TFLOP/s | HBM floor µs | MFMA floor µs | ceiling TFLOP/s | measured / ceiling |')
print('|---|---|---|---|---|---|---|---|---|---|---|')
for tf, kind, m, cin, cout, k, s, cnt, avg, t_h, t_m, ceil in sorted(rows):
    if tf < lim:
        print('| %s | %d | %d→%d | %d/%d | %d | %.1f | %.1f | %.1f | %.1f | %.1f | %.2f |' % (kind, m, cin, cout, k, s, cnt, avg, tf, t_h, t_m, ceil, tf / ceil))
