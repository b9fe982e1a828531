"""Time y3_conv2d_fwd / dgrad / wgrad on the network's main shapes under the tile forced by Y3_TILE (development tool).
    Y3_TILE=128,128,32 python tools/conv_tune.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
import torch                       # noqa: E402
from yolo3 import _hip             # noqa: E402

SHAPES = [  # n, h, w, cin, cout, k, s
    (8, 208, 208, 32, 64, 3, 1),
    (8, 104, 104, 64, 128, 3, 1),
    (8, 52, 52, 128, 256, 3, 1),
    (8, 26, 26, 256, 512, 3, 1),
    (8, 13, 13, 512, 1024, 3, 1),
    (8, 52, 52, 256, 128, 1, 1),
    (8, 26, 26, 512, 256, 1, 1),
    (8, 13, 13, 1024, 512, 1, 1),
]
st = torch.cuda.current_stream().cuda_stream
tile = os.environ.get('Y3_TILE', 'default')
for (n, h, w, cin, cout, k, s) in SHAPES:
    oh, ow = -(-h // s), -(-w // s)
    x = torch.randn(n * h * w * cin, device='cuda')
    y = torch.empty(n * oh * ow * cout, device='cuda')
    wt = torch.randn(k * k * cin * cout, device='cuda') * 0.05
    b = torch.zeros(cout, device='cuda')
    stats = torch.empty(n * oh * ow // 16 * cout + 8192, device='cuda')
    ws = torch.zeros(64 << 20, device='cuda')
    X = _hip.Tensor(x.data_ptr(), n, h, w, cin, cin)
    Y = _hip.Tensor(y.data_ptr(), n, oh, ow, cout, cout)
    res = []
    dw = torch.empty(k * k * cin * cout, device='cuda')
    wws = torch.zeros(int(_hip.lib.y3_conv2d_wgrad_workspace(X, Y, k, s)) // 4 + 16, device='cuda')
    for mode in ('fwd', 'dgrad', 'wgrad'):
        def run():
            if mode == 'fwd':
                return _hip.lib.y3_conv2d_fwd(X, wt.data_ptr(), b.data_ptr(), k, s, Y, _hip.EPI_LRELU, 0.2, None, None, None, stats.data_ptr(), ws.data_ptr(), ws.numel() * 4, st)
            if mode == 'wgrad':
                return _hip.lib.y3_conv2d_wgrad(X, Y, k, s, dw.data_ptr(), wws.data_ptr(), wws.numel() * 4, st)
            return _hip.lib.y3_conv2d_dgrad(Y, wt.data_ptr(), k, s, X, 0, ws.data_ptr(), ws.numel() * 4, st)
        rc = run()
        if rc != 0:
            res.append('%s ERR(%s)' % (mode, _hip.lib.y3_last_error().decode()[:40]))
            continue
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            run()
        e.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(e) / 10 * 1e-3
        res.append('%s %7.1f us %6.1f TF' % (mode, t * 1e6, 2.0 * n * oh * ow * k * k * cin * cout / t / 1e12))
    print('tile %-11s M=%7d cin=%4d cout=%4d k=%d | %s' % (tile, n * oh * ow, cin, cout, k, ' | '.join(res)), flush=True)
