"""Debug: per-layer teacher-forced comparison of the bf16 plan vs the fp64 bf16-emulating oracle."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
sys.path.insert(0, ROOT + '/tests')
import numpy as np
import torch
import test_gpu_model as T

img, n = 96, 2
om, params, yolo, images, _ = T._setup(img, n, 11, True)
yolo.predict(images.cuda(), precision='bf16')
plan = yolo._plan(n, False, True)
torch.cuda.synchronize()
nchw = lambda t: t.torch_view().float().permute(0, 3, 1, 2).cpu()
layers = [nchw(t) for t in plan.layer_out]
ups = [nchw(op[2]) for op in plan.ops if op[0] == 'upsample']
y32 = nchw(plan.tensors[[i for i, t in enumerate(plan.tensors) if t is plan.layer_out[0]][0] + 1])
net = om.Net(params, 3, 2, 2, dtype=torch.float64)
net.bf16 = True
net.trace_exact, net.up_trace = [], []
net.force = {'layers': layers, 'up': ups}
with torch.no_grad():
    net.feature_maps(images.double(), training=False)
net2 = om.Net(params, 3, 2, 2, dtype=torch.float64)
net2.trace = []
with torch.no_grad():
    net2.feature_maps(images.double(), training=False)
exact0 = net2.trace[0]
print('layer0 fp32 kernel vs exact: max abs', float((y32.double() - exact0).abs().max()), 'scale', float(exact0.abs().max()))
for j, (g, r) in enumerate(zip(layers, net.trace_exact)):
    g = g.double()
    ulp = torch.exp2(torch.floor(torch.log2(r.abs().clamp_min(1e-30))) - 7)
    e = (g - r).abs() / ulp
    sc = float(r.abs().max())
    bad = (g - r).abs() > 0.5 * ulp + 3e-5 * sc
    print(j, tuple(r.shape), 'scale %.3g  max err ulps %.3f  nbad %d' % (sc, float(e.max()), int(bad.sum())))
    if j == 0 and bad.any():
        idx = bad.nonzero()[:10]
        for ix in idx:
            ix = tuple(int(v) for v in ix)
            print('   ', ix, 'got', float(g[ix]), 'ref', float(r[ix]), 'y32', float(y32[ix]), 'exact', float(exact0[ix]))
