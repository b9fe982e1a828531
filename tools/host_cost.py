"""Host cost of one launch through the C ABI (ctypes + hipLaunchKernel) and of an event record + wait pair, on tensors small enough
that the GPU keeps up.  Run on the GPU box:  python tools/host_cost.py   (measured: 4.0 us per launch, 8.3 us per event pair)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
import torch
from yolo3 import _hip
lib = _hip.lib
st = torch.cuda.current_stream().cuda_stream
n, h, w, c = 1, 8, 8, 64
a = torch.randn(n*h*w*c, device='cuda'); y = torch.empty_like(a); sc = torch.ones(c, device='cuda'); sf = torch.zeros(c, device='cuda')
A = _hip.Tensor(a.data_ptr(), n, h, w, c, c); Y = _hip.Tensor(y.data_ptr(), n, h, w, c, c)
def t(fn, reps=2000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    dt = time.perf_counter() - t0; torch.cuda.synchronize()
    return dt / reps * 1e6
print('ctypes trivial call (y3_last_error): %.2f us' % t(lambda: lib.y3_last_error()))
print('y3_bn_apply (tiny tensor): %.2f us' % t(lambda: lib.y3_bn_apply(A, sc.data_ptr(), sf.data_ptr(), None, Y, st)))
x = torch.randn(1*16*16*64, device='cuda'); wt = torch.randn(64*64, device='cuda'); b = torch.zeros(64, device='cuda'); o = torch.empty(1*16*16*64, device='cuda')
X = _hip.Tensor(x.data_ptr(), 1, 16, 16, 64, 64); O = _hip.Tensor(o.data_ptr(), 1, 16, 16, 64, 64)
ws = torch.zeros(1 << 20, device='cuda')
print('y3_conv2d_fwd 1x1 (tiny): %.2f us' % t(lambda: lib.y3_conv2d_fwd(X, wt.data_ptr(), b.data_ptr(), 1, 1, O, 1, 0.2, None, None, None, None, ws.data_ptr(), ws.numel()*4, st)))
e = torch.cuda.Event(); s2 = torch.cuda.Stream()
print('event record + wait: %.2f us' % t(lambda: (e.record(), s2.wait_event(e))))
print('torch tiny kernel (a.add_(1)): %.2f us' % t(lambda: a.add_(1)))
