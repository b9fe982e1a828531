import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + '/object-detection-yolov3_amd'); sys.path.insert(0, ROOT + '/tests')
import numpy as np, torch
import torch.nn.functional as F
from yolo3 import _hip as hip
from util import nhwc_buf, stream
from test_gpu_kernels import _conv_ref
for case in [(2, 13, 13, 128, 64, 1, 1), (2, 128, 128, 32, 128, 3, 1), (2, 13, 13, 64, 128, 3, 1)]:
    n, h, w, cin, cout, k, s = case
    g = torch.Generator().manual_seed(1)
    x = torch.randn(n, cin, h, w, generator=g); wk = torch.randn(k, k, cin, cout, generator=g) * 0.1; b = torch.randn(cout, generator=g)
    oh, ow = -(-h // s), -(-w // s)
    sbuf, sv = nhwc_buf(n, h, w, cin, ld=cin + 8, off=4); sv.copy_(x.permute(0, 2, 3, 1))
    old = (cout + 3) // 4 * 4 + 4
    dbuf, dv = nhwc_buf(n, oh, ow, cout, ld=old)
    wd, bd = wk.contiguous().cuda(), b.cuda()
    src = hip.Tensor(sv.data_ptr(), n, h, w, cin, cin + 8); dst = hip.Tensor(dv.data_ptr(), n, oh, ow, cout, old)
    hip.check(hip.lib.y3_conv2d_fwd(src, wd.data_ptr(), bd.data_ptr(), k, s, dst, hip.EPI_LRELU, 0.2, None, None, None, None, None, 0, stream()))
    ref = F.leaky_relu(_conv_ref(x, wk, b, k, s), 0.2).permute(0, 2, 3, 1).reshape(-1, cout).numpy()
    got = dv.cpu().reshape(-1, cout).numpy()
    bad = ~np.isclose(got, ref, rtol=1e-3, atol=1e-3)
    rows = np.where(bad.any(1))[0]; cols = np.where(bad.any(0))[0]
    print(case, 'bad elems', bad.sum(), 'of', bad.size, 'nan', np.isnan(got).sum(), 'rows', rows[:12], '...', rows[-5:] if len(rows) else '', 'cols', cols[:8], len(cols))
