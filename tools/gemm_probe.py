"""How fast is the conv main loop on an ideal GEMM shape (long K, tiles = 4 per CU)?  Development probe."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
import torch
from yolo3 import _hip
st = torch.cuda.current_stream().cuda_stream
for (n, h, w, cin, cout, k) in [(1, 64, 64, 4096, 4096, 1), (1, 64, 64, 1024, 4096, 1), (1, 64, 64, 256, 4096, 1), (2, 64, 64, 512, 2048, 3)]:
    x = torch.randn(n * h * w * cin, device='cuda')
    y = torch.empty(n * h * w * cout, device='cuda')
    wt = torch.randn(k * k * cin * cout, device='cuda') * 0.02
    b = torch.zeros(cout, device='cuda')
    X = _hip.Tensor(x.data_ptr(), n, h, w, cin, cin)
    Y = _hip.Tensor(y.data_ptr(), n, h, w, cout, cout)
    run = lambda: _hip.lib.y3_conv2d_fwd(X, wt.data_ptr(), b.data_ptr(), k, 1, Y, 0, 0.0, None, None, None, None, None, 0, st)
    assert run() == 0, _hip.lib.y3_last_error()
    torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        run()
    e.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(e) / 5 * 1e-3
    print('M=%d K=%d N=%d: %.1f us %.1f TF' % (n * h * w, k * k * cin, cout, t * 1e6, 2.0 * n * h * w * k * k * cin * cout / t / 1e12), flush=True)
