"""Development: ONE x3 forward shape launched N times (a clean target for rocprofv3 --pmc).  python tools/x3_one.py 26 256 512 [iters] [dgrad]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/object-detection-yolov3_amd']
import torch                       # noqa: E402
from yolo3 import _hip             # noqa: E402
hw, cin, cout = (int(sys.argv[i]) for i in (1, 2, 3))
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
lib, st = _hip.lib, torch.cuda.current_stream().cuda_stream
n, k = 8, 3
x = torch.randn(n, hw, hw, cin, device='cuda')
w = (torch.randn(k, k, cout, cin, device='cuda') * 0.05).contiguous()
pl = torch.empty(3 * w.numel(), dtype=torch.bfloat16, device='cuda')
_hip.check(lib.y3_x3_split_weights(w.data_ptr(), pl.data_ptr(), 9, cout, cin, st), 'split')
y = torch.empty(n, hw, hw, cout, device='cuda')
b = torch.zeros(cout, device='cuda')
m = n * hw * hw
wsb = int(lib.y3_conv2d_fwd_workspace_x(m, cin, k, cout, _hip.CONV_X3))
ws = torch.zeros(wsb // 4 + 16, device='cuda')
X, Y = _hip.Tensor(x.data_ptr(), n, hw, hw, cin, cin), _hip.Tensor(y.data_ptr(), n, hw, hw, cout, cout)
for _ in range(iters):
    _hip.check(lib.y3_conv2d_fwd(X, pl.data_ptr(), b.data_ptr(), k, 1, Y, _hip.EPI_LRELU | _hip.CONV_X3, 0.2, None, None, None, None, ws.data_ptr(), wsb, st), 'fwd')
torch.cuda.synchronize()
print('done')
