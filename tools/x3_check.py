"""Development tool: y3_conv2d_fwd / y3_conv2d_dgrad with and without Y3_CONV_X3 on the network's shapes --
error of both against an fp64 reference (torch CPU) and time per launch, alternating launches of the two arithmetics.
    python tools/x3_check.py [--no-ref] [--shapes 52,26,13]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
import torch                       # noqa: E402
import torch.nn.functional as F    # noqa: E402
from yolo3 import _hip             # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--no-ref', action='store_true')
ap.add_argument('--iters', type=int, default=20)
ap.add_argument('--batch', type=int, default=8)
ap.add_argument('--x3-only', action='store_true')
ap.add_argument('--pad', type=int, default=0, help='pixel pitch of the activations = channels + pad floats')
ap.add_argument('--wgrad', action='store_true', help='time the kernel gradient, f32 and x3')
ap.add_argument('--stride2', action='store_true', help='time the five stride-2 forward layers instead')
ap.add_argument('--all', action='store_true', help='every stride-1 shape of the net the x3 kernels take')
args = ap.parse_args()
N = args.batch
SHAPES = [  # n, h, w, cin, cout, k
    (N, 52, 52, 128, 256, 3),
    (N, 26, 26, 256, 512, 3),
    (N, 13, 13, 512, 1024, 3),
]
if args.all:
    SHAPES += [(N, 208, 208, 32, 64, 3), (N, 104, 104, 64, 128, 3), (N, 52, 52, 256, 128, 1), (N, 26, 26, 512, 256, 1), (N, 13, 13, 1024, 512, 1),
               (N, 208, 208, 64, 32, 1), (N, 104, 104, 128, 64, 1), (N, 26, 26, 768, 256, 1), (N, 52, 52, 384, 128, 1)]
lib = _hip.lib
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)


def planes(w):
    """three bf16 piece planes of a CUDA fp32 tensor (the weight operand of a Y3_CONV_X3 forward / data-gradient launch)"""
    w = w.contiguous()                                   # [kh, kw, rows, K per row]
    out = torch.empty(3 * w.numel(), dtype=torch.bfloat16, device=w.device)
    _hip.check(lib.y3_x3_split_weights(w.data_ptr(), out.data_ptr(), w.shape[0] * w.shape[1], w.shape[2], w.shape[3], st), 'split')
    return out

if args.wgrad:
    shapes = [(52, 128, 256, 3, 1), (26, 256, 512, 3, 1), (13, 512, 1024, 3, 1), (104, 64, 128, 3, 1), (104, 128, 256, 3, 2), (52, 256, 512, 3, 2), (26, 512, 1024, 3, 2),
              (52, 256, 128, 1, 1), (26, 512, 256, 1, 1), (13, 1024, 512, 1, 1), (208, 32, 64, 3, 1), (416, 32, 64, 3, 2), (208, 64, 32, 1, 1), (104, 128, 64, 1, 1)]
    for (hh, cin, cout, k, s_) in shapes:
        oh = hh // s_
        xd = torch.randn(N, hh, hh, cin, device='cuda')
        dyd = torch.randn(N, oh, oh, cout, device='cuda')
        X, DY = _hip.Tensor(xd.data_ptr(), N, hh, hh, cin, cin), _hip.Tensor(dyd.data_ptr(), N, oh, oh, cout, cout)
        dws = [torch.empty(k, k, cin, cout, device='cuda') for _ in range(2)]
        wss = []
        for fl in (0, _hip.CONV_X3):
            wsb = int(lib.y3_conv2d_wgrad_workspace_x(X, DY, k, s_, fl))
            wss.append(torch.zeros(wsb // 4 + 16, device='cuda'))
        ok = lib.y3_conv2d_wgrad_x3_ok(N * oh * oh, cin, k, cout)

        def f(x3):
            _hip.check(lib.y3_conv2d_wgrad_x(X, DY, k, s_, dws[x3].data_ptr(), _hip.CONV_X3 if x3 else 0, wss[x3].data_ptr(), wss[x3].numel() * 4, st), 'wgrad')
        evs = []
        for x3 in ((0, 1, 0, 1) if ok else (0, 0)):
            for _ in range(3):
                f(x3)
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.iters):
                f(x3)
            e.record()
            torch.cuda.synchronize()
            evs.append(a.elapsed_time(e) / args.iters * 1e3)
        if ok:
            err = float((dws[0] - dws[1]).abs().max() / dws[0].abs().max())
            print('wgrad %4d->%4d k%d s%d @%3d: f32 %6.1f / %6.1f us   x3 %6.1f / %6.1f us   max |x3 - f32| / max|f32| %.2e' % (cin, cout, k, s_, hh, evs[0], evs[2], evs[1], evs[3], err), flush=True)
        else:
            print('wgrad %4d->%4d k%d s%d @%3d: f32 %6.1f / %6.1f us   (x3 does not take it)' % (cin, cout, k, s_, hh, evs[0], evs[1]), flush=True)
    sys.exit(0)

if args.stride2:
    for (hh, cin, cout) in ((416, 32, 64), (208, 64, 128), (104, 128, 256), (52, 256, 512), (26, 512, 1024)):
        oh = hh // 2
        xd = torch.randn(N, hh, hh, cin, device='cuda')
        wd = torch.randn(3, 3, cin, cout, device='cuda') * 0.05
        wtd = planes(wd.permute(0, 1, 3, 2))
        b = torch.zeros(cout, device='cuda')
        yo = torch.empty(N, oh, oh, cout, device='cuda')
        X, Y = _hip.Tensor(xd.data_ptr(), N, hh, hh, cin, cin), _hip.Tensor(yo.data_ptr(), N, oh, oh, cout, cout)
        m = N * oh * oh
        ws = torch.zeros((256 << 20) // 4, device='cuda')

        def f(x3):
            _hip.check(lib.y3_conv2d_fwd(X, (wtd if x3 else wd).data_ptr(), b.data_ptr(), 3, 2, Y, _hip.EPI_LRELU | (_hip.CONV_X3 if x3 else 0), 0.2,
                                         None, None, None, None, ws.data_ptr(), ws.numel() * 4, st), 'fwd s2')
        evs = []
        for x3 in (0, 1, 0, 1):
            for _ in range(3):
                f(x3)
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(args.iters):
                f(x3)
            e.record()
            torch.cuda.synchronize()
            evs.append(a.elapsed_time(e) / args.iters * 1e3)
        print('stride-2 fwd %4d->%4d @%3d: f32 %6.1f / %6.1f us   x3 %6.1f / %6.1f us' % (cin, cout, hh, evs[0], evs[2], evs[1], evs[3]), flush=True)
    sys.exit(0)


def timeit(fns, iters):
    """alternating launches: [a, b, a, b, ...]; returns ms per launch for each"""
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(iters + 1)] for _ in fns]
    tot = [0.0] * len(fns)
    for _ in range(3):
        for f in fns:
            f()
    torch.cuda.synchronize()
    for it in range(iters):
        for i, f in enumerate(fns):
            a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            f()
            e.record()
            evs[i][it] = (a, e)
    torch.cuda.synchronize()
    for i in range(len(fns)):
        ts = sorted(a.elapsed_time(e) for a, e in evs[i][:iters])
        tot[i] = ts[len(ts) // 2]
    return tot


for (n, h, w, cin, cout, k) in SHAPES:
    taps = k * k
    x = torch.randn(n, cin, h, w, generator=g)
    wk = torch.randn(k, k, cin, cout, generator=g) * (1.0 / (taps * cin) ** 0.5)
    dy = torch.randn(n, cout, h, w, generator=g)
    PAD = args.pad
    xd = torch.zeros(n, h, w, cin + PAD, device='cuda')
    xd[..., :cin] = x.permute(0, 2, 3, 1).cuda()
    dyd = torch.zeros(n, h, w, cout + PAD, device='cuda')
    dyd[..., :cout] = dy.permute(0, 2, 3, 1).cuda()
    wd = wk.contiguous().cuda()                                   # [tap][cin][cout]
    wtd = wk.permute(0, 1, 3, 2).contiguous().cuda()              # [tap][cout][cin]
    pl_fwd, pl_dg = planes(wtd), planes(wd)
    b = torch.zeros(cout, device='cuda')
    yo = [torch.empty(n, h, w, cout + PAD, device='cuda') for _ in range(2)]
    dxo = [torch.empty(n, h, w, cin + PAD, device='cuda') for _ in range(2)]
    X = _hip.Tensor(xd.data_ptr(), n, h, w, cin, cin + PAD)
    DY = _hip.Tensor(dyd.data_ptr(), n, h, w, cout, cout + PAD)
    m = n * h * w
    ws = torch.zeros(max(int(lib.y3_conv2d_fwd_workspace_x(m, cin, k, cout, f)) for f in (0, _hip.CONV_X3)) // 4 + (64 << 20) // 4, device='cuda')
    wsb = ws.numel() * 4

    def fwd(x3):
        Y = _hip.Tensor(yo[x3].data_ptr(), n, h, w, cout, cout + PAD)
        _hip.check(lib.y3_conv2d_fwd(X, (pl_fwd if x3 else wd).data_ptr(), b.data_ptr(), k, 1, Y, _hip.EPI_LRELU | (_hip.CONV_X3 if x3 else 0), 0.2,
                                     None, None, None, None, ws.data_ptr(), wsb, st), 'fwd')

    def dgrad(x3):
        DX = _hip.Tensor(dxo[x3].data_ptr(), n, h, w, cin, cin + PAD)
        _hip.check(lib.y3_conv2d_dgrad(DY, (pl_dg if x3 else wtd).data_ptr(), k, 1, DX, _hip.CONV_X3 if x3 else 0, ws.data_ptr(), wsb, st), 'dgrad')

    ok_f = lib.y3_conv2d_x3_ok(m, cin, taps, cout)
    ok_d = lib.y3_conv2d_x3_ok(m, cout, taps, cin)
    line = 'M=%6d %4d->%4d k%d |' % (m, cin, cout, k)
    flop = 2.0 * m * taps * cin * cout
    if args.x3_only:
        t = timeit([lambda: fwd(1), lambda: dgrad(1)], args.iters)
        line += ' x3 fwd %6.1f us (%5.1f TF)  dgrad %6.1f us (%5.1f TF) |' % (t[0] * 1e3, flop / t[0] / 1e9, t[1] * 1e3, flop / t[1] / 1e9)
    else:
        if ok_f:
            t = timeit([lambda: fwd(0), lambda: fwd(1)], args.iters)
            line += ' fwd f32 %6.1f us (%5.1f TF)  x3 %6.1f us (%5.1f TF) %.2fx |' % (t[0] * 1e3, flop / t[0] / 1e9, t[1] * 1e3, flop / t[1] / 1e9, t[0] / t[1])
        if ok_d:
            t = timeit([lambda: dgrad(0), lambda: dgrad(1)], args.iters)
            line += ' dgrad f32 %6.1f us  x3 %6.1f us %.2fx |' % (t[0] * 1e3, t[1] * 1e3, t[0] / t[1])
    if not args.no_ref:
        nr = min(n, 2)      # fp64 reference on the first images (CPU)
        ref = F.leaky_relu(F.conv2d(x[:nr].double(), wk.double().permute(3, 2, 0, 1), padding=k // 2), 0.2).permute(0, 2, 3, 1)
        for nm, t_ in (('f32', yo[0]), ('x3', yo[1])):
            if nm == 'x3' and not ok_f:
                continue
            err = (t_[:nr, ..., :cout].cpu().double() - ref).abs().max().item() / ref.abs().max().item()
            line += ' fwd %s err %.2e' % (nm, err)
        wflip = wk.double().flip(0, 1).permute(2, 3, 0, 1)          # conv_transpose as a conv with the flipped kernel: [cin][cout][kh][kw]
        refd = F.conv2d(dy[:nr].double(), wflip, padding=k // 2).permute(0, 2, 3, 1)
        for nm, t_ in (('f32', dxo[0]), ('x3', dxo[1])):
            if nm == 'x3' and not ok_d:
                continue
            err = (t_[:nr, ..., :cin].cpu().double() - refd).abs().max().item() / refd.abs().max().item()
            line += ' dgrad %s err %.2e' % (nm, err)
    print(line, flush=True)
