"""Time y3_conv2d_fwd_bf16 on single layer shapes (batch 8 of 608^2 tiles), back-to-back launches.
Tile shapes can be forced with Y3_BF16_TILE=1|2|3 (64x64, 128x128, 256x128).  python tools/bf16_layer.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
import torch  # noqa: E402
from yolo3 import _hip  # noqa: E402

extra = 0
lib = _hip.lib
st = torch.cuda.current_stream().cuda_stream
shapes = [(8, 76, 76, 128, 256, 3, 1), (8, 38, 38, 256, 512, 3, 1), (8, 19, 19, 512, 1024, 3, 1), (8, 76, 76, 256, 128, 1, 1),
          (8, 152, 152, 64, 128, 3, 1), (8, 304, 304, 32, 64, 3, 1), (32, 76, 76, 128, 256, 3, 1)]
for n, h, w, cin, cout, k, s in shapes:
    oh, ow = -(-h // s), -(-w // s)
    x = torch.randn(n, h, w, cin, device='cuda').to(torch.bfloat16)
    wt = (torch.randn(k * k, cout, cin, device='cuda') * 0.05).to(torch.bfloat16)
    b = torch.zeros(cout, device='cuda')
    y = torch.empty(n, oh, ow, cout, device='cuda', dtype=torch.bfloat16)
    src = _hip.Tensor(x.data_ptr(), n, h, w, cin, cin)
    dst = _hip.Tensor(y.data_ptr(), n, oh, ow, cout, cout)
    call = lambda: _hip.check(lib.y3_conv2d_fwd_bf16(src, wt.data_ptr(), b.data_ptr(), k, s, dst, 0, _hip.EPI_LRELU | extra, 0.2, None, None, None, st))
    for _ in range(3):
        call()
    best = 1e9
    for _ in range(5):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            call()
        e.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) / 10 * 1e3)
    fl = 2.0 * n * oh * ow * k * k * cin * cout
    print('N=%d %dx%d %d->%d k%d s%d: %.1f us  %.0f TFLOP/s' % (n, h, w, cin, cout, k, s, best, fl / best / 1e6), flush=True)
