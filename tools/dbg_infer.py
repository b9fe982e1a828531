"""Layer-by-layer comparison of the inference forward against the oracle (debug tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/object-detection-yolov3_amd'); sys.path.insert(0, ROOT + '/tests')
import numpy as np, torch
from oracle import model as om
from yolo3.model import YoloV3
A = [(64, 384), (384, 64)]; K = 2
img, n, seed = int(sys.argv[1]) if len(sys.argv) > 1 else 416, int(sys.argv[2]) if len(sys.argv) > 2 else 1, 7
params = om.init_params(3, 2, K, seed=seed, randomize_bn=True)
for p in params:
    if 'gamma' not in p:
        p['W'] *= 0.02
g = torch.Generator().manual_seed(seed)
images = torch.randn(n, 3, img, img, generator=g)
net = om.Net(params, 3, 2, K, dtype=torch.float64)
net.trace = []
with torch.no_grad():
    fms = net.feature_maps(images.double(), training=False)
yolo = YoloV3(n, [img, img, 3], K, A)
yolo.set_weights(params)
yolo.predict(images.cuda())
plan = yolo._plan(n, False)
torch.cuda.synchronize()
bn_specs = [sp for sp in yolo.specs if sp.bn]
for i, (t, ref, sp) in enumerate(zip(plan.layer_out, net.trace, bn_specs)):
    got = t.torch_view().cpu().double().permute(0, 3, 1, 2)
    err = float((got - ref).abs().max()); sc = float(ref.abs().max())
    flag = '   <<<<' if err > 1e-3 * sc else ''
    print('%2d cin %4d cout %4d k%d s%d  M %6d  scale %.3e err %.3e%s' % (i, sp.cin, sp.cout, sp.k, sp.s, t.m, sc, err, flag))
