"""Back-to-back timing of single bf16 conv layers through the product library:
    python tools/bf16_layer.py [n h cin cout k] ...   (default: the 25-tile shapes of the tiled path)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
import torch  # noqa: E402
from yolo3 import _hip  # noqa: E402
lib = _hip.lib
st = torch.cuda.current_stream().cuda_stream
shapes = [(25, 76, 128, 256, 3), (25, 38, 256, 512, 3), (25, 19, 512, 1024, 3), (25, 76, 256, 128, 1), (25, 38, 512, 256, 1), (25, 19, 1024, 512, 1),
          (8, 76, 128, 256, 3), (8, 38, 256, 512, 3), (8, 19, 512, 1024, 3)]
if len(sys.argv) > 5:
    shapes = [tuple(int(v) for v in sys.argv[1:6])]
for n, h, cin, cout, k in shapes:
    w = h
    x = torch.randn(n, h, w, cin, device='cuda').to(torch.bfloat16)
    wt = (torch.randn(k * k, cout, cin, device='cuda') * 0.05).to(torch.bfloat16)
    b = torch.zeros(cout, device='cuda')
    y = torch.empty(n, h, w, cout, device='cuda', dtype=torch.bfloat16)
    src = _hip.Tensor(x.data_ptr(), n, h, w, cin, cin)
    dst = _hip.Tensor(y.data_ptr(), n, h, w, cout, cout)
    call = lambda: _hip.check(lib.y3_conv2d_fwd_bf16(src, wt.data_ptr(), b.data_ptr(), k, 1, dst, 0, _hip.EPI_LRELU, 0.2, None, None, None, st))
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            call()
        e.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) / 20 * 1e3)
    print('%dx%dx%d %d->%d k%d: %.1f us per launch = %.1f TFLOP/s' % (n, h, w, cin, cout, k, best, 2.0 * n * h * w * k * k * cin * cout / best / 1e6), flush=True)
