"""Per-launch timing of one training step (HIP events around every launch), with achieved TFLOP/s per conv shape.
Run on the GPU box:  python tools/layer_times.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
import numpy as np   # noqa: E402
import torch         # noqa: E402
import bench         # noqa: E402
from yolo3.model import YoloV3   # noqa: E402
from yolo3._hip import check     # noqa: E402

yolo = YoloV3(8, [416, 416, 3], 2, bench.ANCHORS, learning_rate=1e-4, seed=1)
g = torch.Generator().manual_seed(100)
images = torch.randn(8, 3, 416, 416, generator=g).cuda()
gts = [torch.from_numpy(x).cuda() for x in bench.synth_labels(np.random.default_rng(3), 8)]
yolo.train_step((images, gts))
plan = yolo._plan(8, True)
st = torch.cuda.current_stream().cuda_stream
names = {'y3_conv2d_fwd': 'fwd', 'y3_conv2d_dgrad': 'dgrad', 'y3_conv2d_dgrad_bn': 'dgradb', 'y3_conv2d_wgrad': 'wgrad', 'y3_conv2d_wgrad_x': 'wgrad'}
best = {}
for rep in range(3):
    recs = []
    for lst in (plan.fwd, None, plan.bwd):
        if lst is None:
            plan.run_loss(st)
            continue
        for fn, args in lst:
            if fn in ('layer_done', 'record', 'main_wait'):
                continue
            if fn == 'side_call':              # per-launch timing in isolation: everything on one stream here
                fn, args = args[0], args[1]
            nm = names.get(getattr(fn, '__name__', ''))
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            check(fn(*args, st), 'x')
            b.record()
            recs.append((nm or fn.__name__, args, a, b))
    torch.cuda.synchronize()
    for i, (nm, args, a, b) in enumerate(recs):
        best[i] = min(best.get(i, 1e30), a.elapsed_time(b) * 1e3)
tot = {}
rows = []
for i, (nm, args, a, b) in enumerate(recs):
    t = best[i]
    tot[nm] = tot.get(nm, 0) + t
    if nm in ('fwd', 'dgrad', 'dgradb', 'wgrad'):
        if nm == 'fwd':
            src, dst, k, s = args[0], args[5], args[3], args[4]
            m, cin, cout = dst.n * dst.h * dst.w, src.c, dst.c
        elif nm in ('dgrad', 'dgradb'):
            dd, ds, k, s = args[0], args[4], args[2], args[3]
            m, cin, cout = dd.n * dd.h * dd.w, ds.c, dd.c
        else:
            src, dd, k, s = args[0], args[1], args[2], args[3]
            m, cin, cout = dd.n * dd.h * dd.w, src.c, dd.c
        rows.append((nm, m, cin, cout, k, s, t))
agg = {}
for r in rows:
    d = agg.setdefault(r[:6], [0, 0.0, 0.0])
    d[0] += 1
    d[1] += r[6]
    d[2] += 2.0 * r[1] * r[4] * r[4] * r[2] * r[3]
print('%-6s %8s %5s %5s k s  cnt   total_us   avg_us  TFLOP/s' % ('kind', 'M', 'cin', 'cout'))
for key, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-6s %8d %5d %5d %d %d %4d %10.1f %8.1f %8.1f' % (*key, d[0], d[1], d[1] / d[0], d[2] / d[1] / 1e6))
# the streaming kernels, per tensor shape (first argument is the tensor they walk)
other = {}
for i, (nm, args, a, b) in enumerate(recs):
    if nm.startswith('y3_bn') and hasattr(args[0], 'ld'):
        t0 = args[0]
        d = other.setdefault((nm, t0.n * t0.h * t0.w, t0.c), [0, 0.0])
        d[0] += 1
        d[1] += best[i]
for key, d in sorted(other.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
    print('%-20s M=%8d C=%5d cnt %3d total_us %8.1f avg_us %7.1f  %6.0f GB/s of tensor' % (*key, d[0], d[1], d[1] / d[0], key[1] * key[2] * 4 / (d[1] / d[0]) / 1e3))
print('totals (us):', {k: round(v, 1) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])})
print('sum all us', sum(tot.values()))
