"""Fixed (prologue + epilogue + launch) cost vs per-K-step cost of the conv kernel: 1x1 convs of growing Cin at M=86528, N=128."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
import torch
from yolo3 import _hip
st = torch.cuda.current_stream().cuda_stream
n, h, w, cout = 8, 104, 104, 128
import itertools
for mode, (k, cins) in itertools.product(('fwd+stats', 'plain'), ((1, (16, 64, 256, 576, 1024)), (3, (16, 32, 64, 128)))):
    for cin in cins:
        x = torch.randn(n * h * w * cin, device='cuda')
        y = torch.empty(n * h * w * cout, device='cuda')
        wt = torch.randn(k * k * cin * cout, device='cuda') * 0.05
        b = torch.zeros(cout, device='cuda')
        stats = torch.empty(n * h * w // 16 * cout + 8192, device='cuda')
        ws = torch.zeros(64 << 20, device='cuda')
        X = _hip.Tensor(x.data_ptr(), n, h, w, cin, cin)
        Y = _hip.Tensor(y.data_ptr(), n, h, w, cout, cout)
        if mode == 'plain':
            run = lambda: _hip.lib.y3_conv2d_fwd(X, wt.data_ptr(), None, k, 1, Y, 0, 0.0, None, None, None, None, ws.data_ptr(), ws.numel() * 4, st)
        else:
            run = lambda: _hip.lib.y3_conv2d_fwd(X, wt.data_ptr(), b.data_ptr(), k, 1, Y, 1, 0.2, None, None, None, stats.data_ptr(), ws.data_ptr(), ws.numel() * 4, st)
        assert run() == 0, _hip.lib.y3_last_error()
        torch.cuda.synchronize()
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            run()
        e.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(e) / 10 * 1e3
        print('%-9s k=%d K=%4d: %7.1f us   (%.1f TF)' % (mode, k, k * k * cin, t, 2.0 * n * h * w * k * k * cin * cout / t / 1e6), flush=True)
