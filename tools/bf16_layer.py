"""Single-launch vs back-to-back timing of one bf16 conv layer (76x76, 128->256, 3x3, batch 8)."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
import torch  # noqa: E402
from yolo3 import _hip  # noqa: E402
lib = _hip.lib
st = torch.cuda.current_stream().cuda_stream
n, h, w, cin, cout, k = 8, 76, 76, 128, 256, 3
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
for reps in (1, 2, 5, 10, 50):
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize()
        time.sleep(0.01)
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            call()
        e.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) / reps * 1e3)
    print('reps %2d: %.1f us per launch' % (reps, best), flush=True)
# graph of 20 launches
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    stg = torch.cuda.current_stream().cuda_stream
    for _ in range(20):
        _hip.check(lib.y3_conv2d_fwd_bf16(src, wt.data_ptr(), b.data_ptr(), k, 1, dst, 0, _hip.EPI_LRELU, 0.2, None, None, None, stg))
g.replay()
torch.cuda.synchronize()
a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
g.replay()
e.record()
torch.cuda.synchronize()
print('graph of 20: %.1f us per launch' % (a.elapsed_time(e) / 20 * 1e3))
