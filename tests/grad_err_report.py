"""Development helper (not a test): the model-level gradient errors of one training step against the fp64 oracle, per tensor,
for both convolution arithmetics -- to tell sporadic leaky-relu / ignore-mask flips (a few tensors, different ones per arithmetic)
from a systematic error of one arithmetic (many tensors, one of them).   python tests/grad_err_report.py [img] [batch]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'object-detection-yolov3_amd'), os.path.join(ROOT, 'tests')]
import test_gpu_model as T      # noqa: E402

img, n = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (96, 4)
out = {}
for arith in ('f32', 'x3'):
    om, params, yolo, images, gts = T._setup(img, n, 11, False, conv_arithmetic=arith)
    if 'ref' not in out:
        res = {}
        for dt in (torch.float32, torch.float64):
            net = om.Net(params, 3, len(T.ANCHORS), T.K, dtype=dt, requires_grad=True)
            res[dt] = om.train_step(net, om.AdamState(net.trainable(), 1e-3), images.to(dt), [torch.from_numpy(g) for g in gts], (img, img, 3), T.ANCHORS, T.K, n, apply=False)
        out['ref'] = res
    yolo.train_step((images.cuda(), [torch.from_numpy(g).cuda() for g in gts]))
    flat = []
    for sp, d in zip(yolo.specs, yolo.get_gradients()):
        flat += [d['W'], d['b']] + ([d['gamma'], d['beta']] if sp.bn else [])
    out[arith] = [np.asarray(g, np.float64) for g in flat]
r32, r64 = out['ref'][torch.float32]['grads'], out['ref'][torch.float64]['grads']
rows = []
for i, (a, b) in enumerate(zip(r32, r64)):
    a, b = a.numpy().astype(np.float64), b.numpy()
    nb = np.linalg.norm(b) + 1e-30
    rows.append((i, np.linalg.norm(a - b) / nb, np.linalg.norm(out['f32'][i] - b) / nb, np.linalg.norm(out['x3'][i] - b) / nb, np.linalg.norm(out['x3'][i] - out['f32'][i]) / nb, b.size))
e = np.array([[r[1], r[2], r[3], r[4]] for r in rows])
print('tensors %d; rel L2 error vs the fp64 oracle -- median / p90 / max:' % len(rows))
for name, col in (('oracle fp32', 0), ('HIP f32', 1), ('HIP x3', 2), ('HIP x3 vs HIP f32', 3)):
    print('  %-18s %.2e / %.2e / %.2e' % (name, np.median(e[:, col]), np.percentile(e[:, col], 90), e[:, col].max()))
print('ten largest HIP-x3 errors: (tensor, size, oracle fp32, HIP f32, HIP x3)')
for r in sorted(rows, key=lambda r: -r[3])[:10]:
    print('  %3d %8d  %.2e  %.2e  %.2e' % (r[0], r[5], r[1], r[2], r[3]))
print('ten largest HIP-f32 errors:')
for r in sorted(rows, key=lambda r: -r[2])[:10]:
    print('  %3d %8d  %.2e  %.2e  %.2e' % (r[0], r[5], r[1], r[2], r[3]))
