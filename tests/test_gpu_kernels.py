"""GPU parity of every HIP kernel, called through the C ABI (libyolo3hip.so),
against plain torch-CPU fp32/fp64 references of the same op and against the
oracle / golden fixtures.  Tolerances are stated per test."""
import ctypes as C
import glob
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from yolo3 import _hip
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return _hip


def _pad_same(x, k, s):
    from oracle.model import same_pad
    ph, pw = same_pad(x.shape[2], k, s), same_pad(x.shape[3], k, s)
    return F.pad(x, (pw[0], pw[1], ph[0], ph[1]))


def _conv_ref(x, w_k, b, k, s, dtype=torch.float64):
    """x NCHW, w_k Keras [kh,kw,ci,co] -> NCHW, TF SAME padding, in fp64."""
    xx = _pad_same(x.to(dtype), k, s)
    return F.conv2d(xx, w_k.to(dtype).permute(3, 2, 0, 1), None if b is None else b.to(dtype), stride=s)


CONV_CASES = [
    # n, h, w, cin, cout, k, s
    (2, 13, 13, 64, 128, 3, 1),
    (2, 13, 13, 128, 64, 1, 1),
    (2, 26, 26, 32, 64, 3, 2),
    (1, 13, 15, 32, 64, 3, 2),      # odd sizes: SAME pad_before = 1
    (2, 32, 32, 4, 32, 3, 1),       # first layer (channels padded 3 -> 4)
    (2, 13, 13, 256, 14, 1, 1),     # detection head, Cout not a multiple of 4
    (2, 128, 128, 32, 128, 3, 1),   # 256 tiles of 128x128
    (1, 64, 64, 64, 32, 1, 1),      # BN=32 tile
    (3, 20, 20, 64, 64, 3, 1),      # BN=64 tile, M not a tile multiple
    (8, 13, 13, 512, 1024, 3, 1),   # split-K (few tiles, K = 4608)
    (8, 26, 26, 512, 256, 1, 1),    # split-K 1x1
    (8, 13, 13, 256, 512, 3, 2),    # split-K, stride 2, odd input
]


def _x3_or_skip(hip, arith, m, c, taps, nout):
    """flags for the arithmetic under test; skips the x3 variant of shapes the x3 kernels do not take (y3_conv2d_x3_ok)"""
    if arith == 'f32':
        return 0
    if not hip.lib.y3_conv2d_x3_ok(m, c, taps, nout):
        pytest.skip('Y3_CONV_X3 does not take this shape')
    return hip.CONV_X3


def _x3_dgrad_or_skip(hip, arith, dd, k, s, ds):
    """the same for a data gradient (y3_conv2d_dgrad_x3_ok: stride 1 as the forward, stride 2 from 64 input channels up)"""
    if arith == 'f32':
        return 0
    if not hip.lib.y3_conv2d_dgrad_x3_ok(dd, k, s, ds):
        pytest.skip('Y3_CONV_X3 does not take this data gradient')
    return hip.CONV_X3


@pytest.mark.parametrize('arith', ['f32', 'x3'])
@pytest.mark.parametrize('case', CONV_CASES)
def test_conv_fwd(hip, case, arith):
    """y3_conv2d_fwd vs fp64 conv2d; tolerance 2e-5 * max|ref| (fp32 fmaf chain over K <= 1152).  arith = 'x3': the same
    call with Y3_CONV_X3 (three bf16 pieces per operand, weights in the transposed layout), same tolerance."""
    from util import nhwc_buf, stream, assert_close, x3_planes
    n, h, w, cin, cout, k, s = case
    x3 = _x3_or_skip(hip, arith, n * (-(-h // s)) * (-(-w // s)), cin, k * k, cout)
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(n, cin, h, w, generator=g)
    if cin == 4:
        x[:, 3] = 0
    wk = torch.randn(k, k, cin, cout, generator=g) * 0.1
    b = torch.randn(cout, generator=g)
    oh, ow = -(-h // s), -(-w // s)
    sbuf, sv = nhwc_buf(n, h, w, cin, ld=cin + 8, off=4)
    sv.copy_(x.permute(0, 2, 3, 1))
    old = (cout + 3) // 4 * 4 + 4
    dbuf, dv = nhwc_buf(n, oh, ow, cout, ld=old)
    wd, bd = wk.contiguous().cuda(), b.cuda()
    if x3:
        wd = x3_planes(hip, wk.permute(0, 1, 3, 2).contiguous().cuda())      # x3: the three bf16 planes of [kh,kw,co,ci]
    src = hip.Tensor(sv.data_ptr(), n, h, w, cin, cin + 8)
    dst = hip.Tensor(dv.data_ptr(), n, oh, ow, cout, old)
    tiles = hip.lib.y3_conv2d_stats_tiles_x(n * oh * ow, cin, k, cout, x3)
    stats = torch.full((tiles * 2 * cout,), float('nan'), device='cuda')
    wsb = int(hip.lib.y3_conv2d_fwd_workspace_x(n * oh * ow, cin, k, cout, x3))
    ws = torch.zeros(wsb // 4 + 4, device='cuda')        # tickets in the head: zeroed once (yolo3hip.h)
    hip.check(hip.lib.y3_conv2d_fwd(src, wd.data_ptr(), bd.data_ptr(), k, s, dst, hip.EPI_LRELU | x3, 0.2, None, None, None, stats.data_ptr(),
                                    ws.data_ptr(), wsb, stream()))
    ref = F.leaky_relu(_conv_ref(x, wk, b, k, s), 0.2)
    assert_close(dv.cpu().permute(0, 3, 1, 2), ref, rtol=2e-5, what='conv fwd')
    st = stats.view(tiles, 2, cout).double().sum(0).cpu()
    assert_close(st[0], ref.sum(dim=(0, 2, 3)), rtol=1e-4, atol=1e-4 * float(ref.abs().sum(dim=(0, 2, 3)).max()), what='stats sum')
    assert_close(st[1], (ref * ref).sum(dim=(0, 2, 3)), rtol=1e-4, what='stats sumsq')
    # pitch padding untouched
    if old > cout:
        assert torch.isnan(dbuf.view(-1, old)[:, cout:]).all()


@pytest.mark.parametrize('arith', ['f32', 'x3'])
@pytest.mark.parametrize('shape', [(2, 26, 26, 64, 128, 3, 1), (1, 13, 13, 512, 1024, 3, 1), (1, 13, 13, 1024, 512, 1, 1), (1, 26, 26, 1024, 256, 1, 1)])
def test_conv_fwd_fused_inference_epilogue(hip, shape, arith):
    """lrelu -> scale/shift -> + resid (inference-mode BN folded, model.py:38,47); the small-M shapes take the split-K path."""
    from util import nhwc_buf, stream, assert_close, x3_planes
    n, h, w, cin, cout, k, s = shape
    x3 = _x3_or_skip(hip, arith, n * h * w, cin, k * k, cout)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(n, cin, h, w, generator=g)
    wk = torch.randn(k, k, cin, cout, generator=g) * 0.05
    b = torch.randn(cout, generator=g)
    sc = torch.rand(cout, generator=g) + 0.5
    sh = torch.randn(cout, generator=g)
    r = torch.randn(n, cout, h, w, generator=g)
    _, sv = nhwc_buf(n, h, w, cin)
    sv.copy_(x.permute(0, 2, 3, 1))
    _, rv = nhwc_buf(n, h, w, cout, ld=2 * cout, off=cout)
    rv.copy_(r.permute(0, 2, 3, 1))
    _, dv = nhwc_buf(n, h, w, cout)
    wd, bd, scd, shd = wk.contiguous().cuda(), b.cuda(), sc.cuda(), sh.cuda()
    if x3:
        wd = x3_planes(hip, wk.permute(0, 1, 3, 2).contiguous().cuda())
    wsb = int(hip.lib.y3_conv2d_fwd_workspace_x(n * h * w, cin, k, cout, x3))
    ws = torch.zeros(wsb // 4 + 4, device='cuda')        # tickets in the head: zeroed once (yolo3hip.h)
    hip.check(hip.lib.y3_conv2d_fwd(hip.Tensor(sv.data_ptr(), n, h, w, cin, cin), wd.data_ptr(), bd.data_ptr(), k, s,
                                    hip.Tensor(dv.data_ptr(), n, h, w, cout, cout), hip.EPI_LRELU | x3, 0.2, scd.data_ptr(), shd.data_ptr(),
                                    hip.Tensor(rv.data_ptr(), n, h, w, cout, 2 * cout), None, ws.data_ptr(), wsb, stream()))
    ref = F.leaky_relu(_conv_ref(x, wk, b, k, s), 0.2) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None] + r.double()
    assert_close(dv.cpu().permute(0, 3, 1, 2), ref, rtol=2e-5, what='fused epilogue')


BF16_CASES = [
    # n, h, w, cin, cout, k, s, resid, out_f32
    (2, 13, 13, 64, 128, 3, 1, True, False),
    (2, 13, 13, 128, 64, 1, 1, False, False),
    (2, 26, 26, 32, 64, 3, 2, False, False),
    (1, 13, 15, 32, 64, 3, 2, False, False),     # odd sizes, stride 2
    (2, 13, 13, 256, 14, 1, 1, False, True),     # detection head: fp32 out, ragged Cout
    (2, 128, 128, 32, 128, 3, 1, True, False),   # 128x128 tiles
    (1, 64, 64, 64, 32, 1, 1, False, False),     # BN=32 tile
    (4, 128, 128, 32, 128, 1, 1, False, False),  # 512 tiles of 128x128 WITHOUT a residual: the staged epilogue (round 4; the direct one before)
    (4, 100, 100, 64, 128, 1, 1, False, False),  # 157 tiles of 256x128 (K steps of 64) without a residual: staged epilogue too
    (3, 20, 20, 64, 64, 3, 1, True, False),      # BN=64 tile, ragged M
    (1, 13, 13, 512, 1024, 3, 1, True, False),   # long K
    (1, 26, 26, 768, 256, 1, 1, False, False),   # 1x1 with non-power-of-two Cin (route concat)
    (8, 13, 13, 512, 1024, 3, 1, True, False),   # split-K: 352 tiles x 144 K steps -> 6 slices
    (2, 19, 19, 256, 512, 3, 2, False, False),   # split-K, stride 2, ragged M (200 rows)
    (8, 26, 26, 1024, 256, 1, 1, False, True),   # split-K on a 1x1 (32 K steps -> 2 slices), fp32 out
    # the 256 x 256 ping-pong kernel (conv_bf16_pp_kernel: Cin % 64 == 0, Cout >= 256, >= 96 tiles)
    (1, 78, 79, 64, 1024, 1, 1, False, False),   # ONE K tile (+ the all-zero pad tile), ragged M (6162 rows)
    (1, 80, 80, 64, 1024, 3, 1, True, False),    # 3x3 zero padding through the DMA masks, 9 K tiles (odd), residual
    (2, 80, 80, 64, 320, 1, 1, False, True),     # Cout 320: second column tile a quarter full, fp32 out
    (1, 160, 160, 64, 1024, 3, 2, False, False), # stride 2 (asymmetric SAME padding)
    (1, 80, 80, 128, 1024, 1, 1, True, False),   # 2 K tiles, residual
    (1, 78, 80, 192, 1024, 1, 1, False, False),  # non-power-of-two Cin (1x1), 3 K tiles
    # the patch kernel (conv_bf16_c32_kernel: 3x3, Cin 32 -> Cout 64; the two stride-2 cases near the top run on it too)
    (2, 40, 45, 32, 64, 3, 1, True, False),      # stride 1, residual, ragged strip (45 = 32 + 13) and row group (40 = 5 x 8)
    (1, 13, 70, 32, 64, 3, 1, False, False),     # rows not a multiple of the group, three strips
    (6, 256, 256, 32, 64, 3, 1, True, False),    # 1 536 groups on 512 workgroups: three groups each (both LDS buffers re-used)
    (2, 67, 131, 32, 64, 3, 2, False, False),    # stride 2, odd sizes (SAME padding before = 1), three strips
    (5, 256, 256, 32, 64, 3, 2, False, False),   # stride 2, even sizes (padding after only), 640 groups on 512 workgroups
    (2, 30, 40, 32, 64, 3, 2, True, False),      # stride 2 with a residual (the fourth instantiation)
    # conv_bf16_c64_kernel (3x3 stride 1, Cin 64 -> Cout 128: two K halves per channel block, exchanged through LDS)
    (2, 40, 45, 64, 128, 3, 1, True, False),     # ragged strip, residual
    (1, 9, 70, 64, 128, 3, 1, False, False),     # rows not a multiple of the group (9 = 2 x 4 + 1), three strips
    (3, 152, 152, 64, 128, 3, 1, True, False),   # 570 groups on 256 workgroups: both patch buffers and the exchange buffer re-used
    (2, 67, 131, 64, 128, 3, 2, False, False),   # stride 2 (two planes of 128-byte pixels), odd sizes
    (2, 160, 160, 64, 128, 3, 2, False, False),  # stride 2, even sizes, 480 groups on 256 workgroups
]


@pytest.mark.parametrize('case', BF16_CASES)
def test_conv_fwd_bf16(hip, case):
    """y3_conv2d_fwd_bf16 vs an fp64 conv over the SAME bf16-rounded operands: products of bf16 pairs are exact in
    fp32, so the only differences are fp32 accumulation order (<= 2e-5 * max|ref|, as for the fp32 kernel) and, for
    bf16 outputs, one round-to-nearest-even (half an ulp = 2^-9 relative)."""
    from util import stream, assert_close
    n, h, w, cin, cout, k, s, with_resid, out_f32 = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    bf = lambda t: t.to(torch.bfloat16)
    x = bf(torch.randn(n, cin, h, w, generator=g))
    wk = bf(torch.randn(k, k, cin, cout, generator=g) * 0.1)
    b = torch.randn(cout, generator=g)
    sc = torch.rand(cout, generator=g) + 0.5
    sh = torch.randn(cout, generator=g)
    oh, ow = -(-h // s), -(-w // s)
    r = bf(torch.randn(n, cout, oh, ow, generator=g))
    sld, dld = cin + 8, (cout + 7) // 8 * 8 + 8
    sbuf = torch.full((n * h * w * sld + 8,), float('nan'), dtype=torch.bfloat16, device='cuda')
    sv = torch.as_strided(sbuf, (n, h, w, cin), (h * w * sld, w * sld, sld, 1), 8)
    sv.copy_(x.permute(0, 2, 3, 1))
    ddt = torch.float32 if out_f32 else torch.bfloat16
    dbuf = torch.full((n * oh * ow * dld,), float('nan'), dtype=ddt, device='cuda')
    dv = dbuf.view(n, oh, ow, dld)[..., :cout]
    rbuf = r.permute(0, 2, 3, 1).contiguous().cuda()
    # weights: Keras [kh,kw,ci,co] fp32 -> [kh,kw,co,ci] -> bf16, through the library's own kernels
    wf = wk.float().contiguous().cuda()
    wt = torch.empty(k * k * cout * cin, device='cuda')
    hip.check(hip.lib.y3_transpose_weights(wf.data_ptr(), wt.data_ptr(), k * k, cin, cout, stream()))
    wtb = torch.empty(k * k * cout * cin, dtype=torch.bfloat16, device='cuda')
    hip.check(hip.lib.y3_f32_to_bf16(wt.data_ptr(), wtb.data_ptr(), wt.numel(), stream()))
    assert torch.equal(wtb.view(k, k, cout, cin).cpu(), wk.permute(0, 1, 3, 2))
    bd, scd, shd = b.cuda(), sc.cuda(), sh.cuda()
    # through the workspace entry: the small-M cases are then split along K (in-kernel ticketed reduction); run twice on the
    # same workspace -- the tickets must be left at zero -- and once more without a workspace (whole-K path): same tolerance
    wsb = int(hip.lib.y3_conv2d_fwd_bf16_workspace(n * oh * ow, cin, k, cout))
    ws = torch.zeros(wsb // 4 + 4, device='cuda')
    for rep in range(2):
        dbuf.fill_(float('nan'))
        hip.check(hip.lib.y3_conv2d_fwd_bf16_ws(hip.Tensor(sv.data_ptr(), n, h, w, cin, sld), wtb.data_ptr(), bd.data_ptr(), k, s,
                                                hip.Tensor(dv.data_ptr(), n, oh, ow, cout, dld), int(out_f32), hip.EPI_LRELU, 0.2,
                                                scd.data_ptr(), shd.data_ptr(),
                                                hip.Tensor(rbuf.data_ptr(), n, oh, ow, cout, cout) if with_resid else None,
                                                ws.data_ptr(), wsb, stream()))
        if rep == 0:
            first_ws = dbuf.clone()
    assert torch.equal(torch.nan_to_num(dbuf.float(), nan=-7.0), torch.nan_to_num(first_ws.float(), nan=-7.0)), 'split-K launch not repeatable on one workspace'
    if wsb:
        assert int(ws[:16384].view(torch.int32).abs().sum()) == 0, 'tickets not reset'
    ref = F.leaky_relu(_conv_ref(x.float(), wk.float(), b, k, s), 0.2) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
    if with_resid:
        ref = ref + r.double()
    got = dv.float().cpu().permute(0, 3, 1, 2)
    if out_f32:
        assert_close(got, ref, rtol=2e-5, what='bf16 conv (fp32 out)')
    else:
        scale = float(ref.abs().max())
        err = (got.double() - ref).abs()
        bound = ref.abs() * 2.0 ** -8 + 2e-5 * scale     # half-ulp rounding (2^-9, doubled for slack) + accumulation order
        assert torch.isfinite(got).all() and bool((err <= bound).all()), 'bf16 conv: max excess %.3e' % float((err - bound).max())
    assert torch.isnan(dbuf.view(-1, dld)[:, cout:].float()).all(), 'pitch padding overwritten'


def test_conv_bf16_pingpong_kernel_is_race_free_at_benchmark_shape(hip):
    """conv_bf16_pp_kernel at a shape of the tiled path (8 x 76 x 76, 128 -> 256, 3x3: 181 workgroups, 18 K tiles): 40
    launches must give the same bits (its LDS-DMA ring is ordered only by counted vmcnt waits and barriers: a read placed a
    phase too early would return stale LDS bytes whenever the DMA happens to land late), and those bits must agree with the
    fp32 MFMA kernel run on the same bf16-representable operands (exact products; fp32 accumulation order + one bf16
    rounding: half an ulp + 2e-5 of scale)."""
    from util import stream
    n, h, w, cin, cout, k = 8, 76, 76, 128, 256, 3
    g = torch.Generator().manual_seed(77)
    x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16)
    wk = (torch.randn(k, k, cin, cout, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(cout, generator=g)
    xd = x.cuda().contiguous()
    wf = wk.float().contiguous().cuda()
    wt = torch.empty(k * k * cout * cin, device='cuda')
    hip.check(hip.lib.y3_transpose_weights(wf.data_ptr(), wt.data_ptr(), k * k, cin, cout, stream()))
    wtb = torch.empty(k * k * cout * cin, dtype=torch.bfloat16, device='cuda')
    hip.check(hip.lib.y3_f32_to_bf16(wt.data_ptr(), wtb.data_ptr(), wt.numel(), stream()))
    bd = b.cuda()
    src = hip.Tensor(xd.data_ptr(), n, h, w, cin, cin)
    outs = []
    y = torch.empty(n, h, w, cout, dtype=torch.bfloat16, device='cuda')
    for rep in range(40):
        y.fill_(float('nan'))
        hip.check(hip.lib.y3_conv2d_fwd_bf16(src, wtb.data_ptr(), bd.data_ptr(), k, 1, hip.Tensor(y.data_ptr(), n, h, w, cout, cout), 0,
                                             hip.EPI_LRELU, 0.2, None, None, None, stream()))
        if rep == 0:
            first = y.clone()
        else:
            assert torch.equal(y.view(torch.int16), first.view(torch.int16)), 'launch %d differs from launch 0' % rep
    # fp32 MFMA kernel on the same operands
    x32 = xd.float().contiguous()
    y32 = torch.empty(n, h, w, cout, device='cuda')
    wsb = int(hip.lib.y3_conv2d_fwd_workspace(n * h * w, cin, k, cout))
    ws = torch.zeros(wsb // 4 + 4, device='cuda')
    hip.check(hip.lib.y3_conv2d_fwd(hip.Tensor(x32.data_ptr(), n, h, w, cin, cin), wf.data_ptr(), bd.data_ptr(), k, 1,
                                    hip.Tensor(y32.data_ptr(), n, h, w, cout, cout), hip.EPI_LRELU, 0.2, None, None, None, None, ws.data_ptr(), wsb, stream()))
    ref, got = y32.double(), first.double()
    scale = float(ref.abs().max())
    bound = ref.abs() * 2.0 ** -8 + 2e-5 * scale
    assert bool(torch.isfinite(got).all()) and bool(((got - ref).abs() <= bound).all()), 'max excess %.3e' % float(((got - ref).abs() - bound).max())


@pytest.mark.parametrize('shape', [(8, 304, 304, 32, 64, 1), (4, 608, 608, 32, 64, 2), (8, 152, 152, 64, 128, 1), (6, 304, 304, 64, 128, 2)])
def test_conv_bf16_patch_kernels_are_race_free_at_tiled_shapes(hip, shape):
    """conv_bf16_c32_kernel / conv_bf16_c64_kernel at the shapes of the tiled path's 608^2 -> 152^2 stages (a few tiles: every
    workgroup walks 5-12 row groups, so both LDS patch buffers -- and, for c64, the accumulator exchange buffer -- are re-used
    many times between barriers): 20 launches must give the same bits, with a residual at stride 1 as in the network, and those
    bits must agree with the fp32 MFMA kernel on the same bf16-representable operands (exact products; fp32 accumulation order
    + one bf16 rounding: half an ulp + 2e-5 of scale)."""
    from util import stream
    n, h, w, cin, cout, s = shape
    k = 3
    oh, ow = -(-h // s), -(-w // s)
    g = torch.Generator().manual_seed(5 + cin + s)
    xd = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16).cuda()
    wk = (torch.randn(k, k, cin, cout, generator=g) * 0.08).to(torch.bfloat16)
    bd = torch.randn(cout, generator=g).cuda()
    rd = torch.randn(n, oh, ow, cout, generator=g).to(torch.bfloat16).cuda() if s == 1 else None
    wf = wk.float().contiguous().cuda()
    wt = torch.empty(k * k * cout * cin, device='cuda')
    hip.check(hip.lib.y3_transpose_weights(wf.data_ptr(), wt.data_ptr(), k * k, cin, cout, stream()))
    wtb = torch.empty(k * k * cout * cin, dtype=torch.bfloat16, device='cuda')
    hip.check(hip.lib.y3_f32_to_bf16(wt.data_ptr(), wtb.data_ptr(), wt.numel(), stream()))
    src = hip.Tensor(xd.data_ptr(), n, h, w, cin, cin)
    res = hip.Tensor(rd.data_ptr(), n, oh, ow, cout, cout) if rd is not None else None
    y = torch.empty(n, oh, ow, cout, dtype=torch.bfloat16, device='cuda')
    for rep in range(20):
        y.fill_(float('nan'))
        hip.check(hip.lib.y3_conv2d_fwd_bf16(src, wtb.data_ptr(), bd.data_ptr(), k, s, hip.Tensor(y.data_ptr(), n, oh, ow, cout, cout), 0,
                                             hip.EPI_LRELU, 0.2, None, None, res, stream()))
        if rep == 0:
            first = y.clone()
        else:
            assert torch.equal(y.view(torch.int16), first.view(torch.int16)), 'launch %d differs from launch 0' % rep
    x32 = xd.float().contiguous()
    y32 = torch.empty(n, oh, ow, cout, device='cuda')
    wsb = int(hip.lib.y3_conv2d_fwd_workspace(n * oh * ow, cin, k, cout))
    ws = torch.zeros(wsb // 4 + 4, device='cuda')
    hip.check(hip.lib.y3_conv2d_fwd(hip.Tensor(x32.data_ptr(), n, h, w, cin, cin), wf.data_ptr(), bd.data_ptr(), k, s,
                                    hip.Tensor(y32.data_ptr(), n, oh, ow, cout, cout), hip.EPI_LRELU, 0.2, None, None, None, None, ws.data_ptr(), wsb, stream()))
    ref = y32.double() + (rd.double() if rd is not None else 0.0)
    got = first.double()
    scale = float(ref.abs().max())
    bound = ref.abs() * 2.0 ** -8 + 2e-5 * scale
    assert bool(torch.isfinite(got).all()) and bool(((got - ref).abs() <= bound).all()), 'max excess %.3e' % float(((got - ref).abs() - bound).max())


def test_conv_bf16_no_patch_flag_falls_back_to_the_ring_kernel(hip):
    """Y3_BF16_NO_PATCH keeps a 32 -> 64 / 64 -> 128 3x3 launch off the patch kernels (yolo3/model.py sets it for layers that
    move less than 300 MB): same operands, both kernels accumulate in fp32 and round once, so the two outputs agree to one bf16
    ulp + accumulation order.  (For 32 -> 64 they are bit-identical: the patch kernel walks K in the ring kernel's order -- tap,
    then 16-channel step -- on the same instruction; the 64 -> 128 kernel adds its two K halves at the end.)"""
    from util import stream
    for cin, cout, s in ((32, 64, 1), (64, 128, 2)):
        n, h, w, k = 2, 64, 96, 3
        oh, ow = -(-h // s), -(-w // s)
        g = torch.Generator().manual_seed(31 + cin)
        xd = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16).cuda()
        wk = (torch.randn(k, k, cin, cout, generator=g) * 0.08).to(torch.bfloat16)
        bd = torch.randn(cout, generator=g).cuda()
        wf = wk.float().contiguous().cuda()
        wt = torch.empty(k * k * cout * cin, device='cuda')
        hip.check(hip.lib.y3_transpose_weights(wf.data_ptr(), wt.data_ptr(), k * k, cin, cout, stream()))
        wtb = torch.empty(k * k * cout * cin, dtype=torch.bfloat16, device='cuda')
        hip.check(hip.lib.y3_f32_to_bf16(wt.data_ptr(), wtb.data_ptr(), wt.numel(), stream()))
        outs = []
        for flags in (hip.EPI_LRELU, hip.EPI_LRELU | hip.BF16_NO_PATCH):
            y = torch.full((n, oh, ow, cout), float('nan'), dtype=torch.bfloat16, device='cuda')
            hip.check(hip.lib.y3_conv2d_fwd_bf16(hip.Tensor(xd.data_ptr(), n, h, w, cin, cin), wtb.data_ptr(), bd.data_ptr(), k, s,
                                                 hip.Tensor(y.data_ptr(), n, oh, ow, cout, cout), 0, flags, 0.2, None, None, None, stream()))
            outs.append(y.double())
        a, b = outs
        scale = float(a.abs().max())
        assert bool(torch.isfinite(b).all()) and bool(((a - b).abs() <= a.abs() * 2.0 ** -7 + 4e-5 * scale).all())


def test_conv_first_bf16(hip):
    """y3_conv2d_first_bf16 (direct fp32 convolution of the RGB layer, bf16 store) vs fp64: half a bf16 ulp + 1e-5 of scale."""
    from util import nhwc_buf, stream
    n, h, w = 2, 37, 44
    g = torch.Generator().manual_seed(17)
    x = torch.randn(n, 4, h, w, generator=g)
    x[:, 3] = 0
    wk = torch.randn(3, 3, 4, 32, generator=g) * 0.3
    b, sc, sh = torch.randn(32, generator=g), torch.rand(32, generator=g) + 0.5, torch.randn(32, generator=g)
    _, sv = nhwc_buf(n, h, w, 4)
    sv.copy_(x.permute(0, 2, 3, 1))
    dbuf = torch.full((n * h * w * 40,), float('nan'), dtype=torch.bfloat16, device='cuda')
    wd, bd, scd, shd = wk.contiguous().cuda(), b.cuda(), sc.cuda(), sh.cuda()
    hip.check(hip.lib.y3_conv2d_first_bf16(hip.Tensor(sv.data_ptr(), n, h, w, 4, 4), wd.data_ptr(), bd.data_ptr(),
                                           hip.Tensor(dbuf.data_ptr(), n, h, w, 32, 40), hip.EPI_LRELU, 0.2, scd.data_ptr(), shd.data_ptr(), stream()))
    ref = (F.leaky_relu(_conv_ref(x, wk, b, 3, 1), 0.2) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]).permute(0, 2, 3, 1)
    got = dbuf.view(n, h, w, 40)[..., :32].double().cpu()
    ulp = torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 7)
    assert bool(((got - ref).abs() <= 0.5 * ulp + 1e-5 * float(ref.abs().max())).all())
    assert torch.isnan(dbuf.view(n, h, w, 40)[..., 32:].float()).all()
    with pytest.raises(hip.HipError):
        hip.check(hip.lib.y3_conv2d_first_bf16(hip.Tensor(sv.data_ptr(), n, h, w, 4, 4), wd.data_ptr(), bd.data_ptr(),
                                               hip.Tensor(dbuf.data_ptr(), n, h, w, 16, 40), 0, 0.0, None, None, stream()))


def test_upsample_and_convert_bf16(hip):
    from util import stream
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 5, 7, 128, generator=g)
    xb = x.to(torch.bfloat16).cuda()
    cv = torch.empty(x.numel(), dtype=torch.bfloat16, device='cuda')
    hip.check(hip.lib.y3_f32_to_bf16(x.cuda().data_ptr(), cv.data_ptr(), x.numel(), stream()))
    assert torch.equal(cv.view_as(xb), xb)                    # round-to-nearest-even, same as torch
    out = torch.full((2, 10, 14, 384), float('nan'), dtype=torch.bfloat16, device='cuda')
    hip.check(hip.lib.y3_upsample_sum2x_fwd_bf16(hip.Tensor(xb.data_ptr(), 2, 5, 7, 128, 128), hip.Tensor(out.data_ptr(), 2, 10, 14, 128, 384), stream()))
    ref = xb.double().sum(-1)
    got = out[..., :128].double()
    assert torch.isnan(out[..., 128:].float()).all()
    up = ref.repeat_interleave(2, 1).repeat_interleave(2, 2)[..., None].expand(-1, -1, -1, 128)
    assert bool(((got - up).abs() <= up.abs() * 2.0 ** -8 + 1e-5).all())


@pytest.mark.parametrize('shape', [(3, 13, 13, 1024, 14), (2, 26, 26, 512, 14), (8, 52, 52, 256, 14), (1, 7, 9, 64, 16), (2, 13, 13, 256, 7)])
def test_conv_fwd_detection_head(hip, shape):
    """detection_layer (model.py:108-120): linear 1x1 conv to <= 16 channels (ragged Cout, pitch 16, no stats / BN fold /
    residual), with and without the activation; fp64 reference, 2e-5 of max|ref|."""
    from util import nhwc_buf, stream, assert_close
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(cin + cout)
    x = torch.randn(n, cin, h, w, generator=g)
    wk = torch.randn(1, 1, cin, cout, generator=g) * 0.05
    b = torch.randn(cout, generator=g)
    _, sv = nhwc_buf(n, h, w, cin)
    sv.copy_(x.permute(0, 2, 3, 1))
    dld = 16
    dbuf, dv = nhwc_buf(n, h, w, cout, ld=dld)
    wd, bd = wk.contiguous().cuda(), b.cuda()
    wsb = int(hip.lib.y3_conv2d_fwd_workspace(n * h * w, cin, 1, cout))      # the heads split K: few output columns, long K
    ws = torch.zeros(max(wsb, 16), dtype=torch.uint8, device='cuda')
    for flags, ref in ((0, _conv_ref(x, wk, b, 1, 1)), (hip.EPI_LRELU, F.leaky_relu(_conv_ref(x, wk, b, 1, 1), 0.2))):
        dbuf.fill_(float('nan'))
        hip.check(hip.lib.y3_conv2d_fwd(hip.Tensor(sv.data_ptr(), n, h, w, cin, cin), wd.data_ptr(), bd.data_ptr(), 1, 1,
                                        hip.Tensor(dv.data_ptr(), n, h, w, cout, dld), flags, 0.2, None, None, None, None, ws.data_ptr(), wsb, stream()))
        assert_close(dv.cpu().permute(0, 3, 1, 2), ref, rtol=2e-5, what='head conv')
        if dld > cout:
            assert torch.isnan(dbuf.view(-1, dld)[:, cout:]).all()


DGRAD_CASES = [
    (8, 13, 13, 512, 1024, 3, 1),   # split-K
    (8, 26, 26, 256, 512, 3, 2),    # split-K across the four parity launches
    (2, 13, 13, 64, 128, 3, 1),
    (2, 13, 13, 128, 64, 1, 1),
    (2, 26, 26, 32, 64, 3, 2),
    (1, 13, 15, 32, 64, 3, 2),
    (2, 13, 13, 256, 14, 1, 1),
    (2, 64, 64, 64, 128, 3, 2),
    (1, 27, 31, 64, 64, 3, 2),      # odd sizes: the four parity classes have different pixel counts; 64-column tiles of the x3 merged launch
    (3, 30, 26, 128, 256, 3, 2),    # x3 merged launch with per-class K slices, row tiles not full
]


@pytest.mark.parametrize('arith', ['f32', 'x3'])
@pytest.mark.parametrize('case', DGRAD_CASES)
@pytest.mark.parametrize('accum', [False, True])
def test_conv_dgrad(hip, case, accum, arith):
    """y3_conv2d_dgrad vs autograd of the fp64 conv; 2e-5 * max|ref|.  arith = 'x3': Y3_CONV_X3 (stride 1, and the merged stride-2 launch
    from 64 input channels up), weights as planes of the Keras layout."""
    from util import nhwc_buf, stream, assert_close
    n, h, w, cin, cout, k, s = case
    g = torch.Generator().manual_seed(11)
    x = torch.randn(n, cin, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    wk = torch.randn(k, k, cin, cout, generator=g) * 0.1
    y = _conv_ref(x, wk, None, k, s)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    oh, ow = y.shape[2], y.shape[3]
    cld = (cout + 3) // 4 * 4
    _, ddv = nhwc_buf(n, oh, ow, cout, ld=cld, fill=float('nan'))
    ddv.copy_(dy.permute(0, 2, 3, 1))
    _, dsv = nhwc_buf(n, h, w, cin, ld=cin + 4, fill=0.0)
    init = torch.randn(n, h, w, cin, generator=g)
    if accum:
        dsv.copy_(init)
    wt = wk.permute(0, 1, 3, 2).contiguous().cuda()          # [kh,kw,co,ci]
    # also exercise y3_transpose_weights
    wt2 = torch.empty_like(wt)
    wd = wk.contiguous().cuda()
    hip.check(hip.lib.y3_transpose_weights(wd.data_ptr(), wt2.data_ptr(), k * k, cin, cout, stream()))
    assert torch.equal(wt, wt2)
    DD, DS = hip.Tensor(ddv.data_ptr(), n, oh, ow, cout, cld), hip.Tensor(dsv.data_ptr(), n, h, w, cin, cin + 4)
    x3 = _x3_dgrad_or_skip(hip, arith, DD, k, s, DS)
    wsb = int(hip.lib.y3_conv2d_dgrad_workspace_x(DD, k, s, DS, x3))
    ws = torch.zeros(wsb // 4 + 4, device='cuda')        # tickets in the head: zeroed once (yolo3hip.h)
    from util import x3_planes
    wop = x3_planes(hip, wd) if x3 else wt2              # x3: the planes of the Keras layout
    hip.check(hip.lib.y3_conv2d_dgrad(DD, wop.data_ptr(), k, s, DS, (hip.EPI_ACCUM if accum else 0) | x3, ws.data_ptr(), wsb, stream()))
    ref = x.grad.permute(0, 2, 3, 1)
    if accum:
        ref = ref + init.double()
    assert_close(dsv.cpu(), ref, rtol=2e-5, what='dgrad')


@pytest.mark.parametrize('shape', [(8, 13, 13, 512, 1024, 3), (8, 26, 26, 128, 256, 1), (2, 52, 52, 64, 128, 3), (8, 52, 52, 128, 256, 3), (1, 13, 15, 64, 32, 1),
                                   (2, 104, 104, 64, 128, 3, 2), (8, 26, 26, 256, 512, 3, 2), (1, 30, 26, 32, 64, 3, 2), (2, 27, 31, 128, 128, 3, 2)])
@pytest.mark.parametrize('arith', ['f32', 'x3'])
@pytest.mark.parametrize('accum', [False, True])
def test_conv_dgrad_bn_epilogue_stats(hip, shape, accum, arith):
    """y3_conv2d_dgrad_bn: the data gradient is bit-identical to y3_conv2d_dgrad's, and the per-row-tile partial moments
    it leaves behind, through y3_bn_bwd_finalize_tiles, give the dgamma / dbeta / dbias / coefficients that y3_bn_bwd_stats
    computes from the finished gradient (fp32 tile sums vs fp64 running sums: 2e-5 of the largest value per quantity;
    dbias is a difference of such sums, compared at 1e-4 of sum|dz| like test_batchnorm_train_fwd_bwd)."""
    from util import nhwc_buf, stream, assert_close
    if os.environ.get('Y3_NO_FAST'):
        pytest.skip('the epilogue statistics live in the fast kernel only (Y3_NO_FAST is set: y3_conv2d_dgrad_bn_tiles() == 0, the model falls back to y3_bn_bwd_stats)')
    n, h, w, cin, cout, k = shape[:6]      # conv cin -> cout; its data gradient has cin channels (h, w: the conv's INPUT size)
    s_ = shape[6] if len(shape) > 6 else 1   # stride 2: the merged launch of the four parity classes carries the statistics
    oh, ow = -(-h // s_), -(-w // s_)
    x3 = _x3_dgrad_or_skip(hip, arith, hip.Tensor(0, n, oh, ow, cout, cout), k, s_, hip.Tensor(0, n, h, w, cin, cin + 4))
    g = torch.Generator().manual_seed(cin * 3 + cout + k)
    dy = torch.randn(n, oh, ow, cout, generator=g)
    wk = torch.randn(k, k, cin, cout, generator=g) * 0.05
    a = torch.randn(n, h, w, cin, generator=g)            # activation of the layer that produced the conv's input
    a = torch.where(a > 0, a, 0.2 * a)
    init = torch.randn(n, h, w, cin, generator=g)
    _, ddv = nhwc_buf(n, oh, ow, cout)
    ddv.copy_(dy)
    _, av = nhwc_buf(n, h, w, cin, ld=cin + 8)
    av.copy_(a)
    wt = (wk if x3 else wk.permute(0, 1, 3, 2)).contiguous().cuda()      # x3: the Keras layout, K (= cout) contiguous per column ...
    if x3:
        from util import x3_planes
        wt = x3_planes(hip, wt)                                           # ... as three bf16 planes
    DD, A = hip.Tensor(ddv.data_ptr(), n, oh, ow, cout, cout), hip.Tensor(av.data_ptr(), n, h, w, cin, cin + 8)
    outs = []
    for fused in (False, True):
        _, dsv = nhwc_buf(n, h, w, cin, ld=cin + 4, fill=0.0)
        if accum:
            dsv.copy_(init)
        DS = hip.Tensor(dsv.data_ptr(), n, h, w, cin, cin + 4)
        wsb = int(hip.lib.y3_conv2d_dgrad_workspace_x(DD, k, s_, DS, x3))
        ws = torch.zeros(wsb // 4 + 4, device='cuda')
        flags = (hip.EPI_ACCUM if accum else 0) | x3
        if fused:
            tiles = int(hip.lib.y3_conv2d_dgrad_bn_tiles_x(DD, k, s_, DS, x3))
            assert tiles > 0
            part = torch.full((tiles * 6 * cin,), float('nan'), device='cuda')
            hip.check(hip.lib.y3_conv2d_dgrad_bn(DD, wt.data_ptr(), k, s_, DS, flags, A, part.data_ptr(), ws.data_ptr(), wsb, stream()))
            assert not torch.isnan(part).any()
        else:
            hip.check(hip.lib.y3_conv2d_dgrad(DD, wt.data_ptr(), k, s_, DS, flags, ws.data_ptr(), wsb, stream()))
        outs.append(dsv.clone().contiguous())
    assert torch.equal(outs[0], outs[1]), 'dgrad_bn changed the data gradient'
    M = n * h * w
    gd, mean, rstd = torch.rand(cin, generator=g).cuda() + 0.5, torch.randn(cin, generator=g).cuda() * 0.1, torch.rand(cin, generator=g).cuda() + 0.5
    ref = [torch.empty(cin, device='cuda') for _ in range(3)] + [torch.empty(3 * cin, device='cuda')]
    got = [torch.empty(cin, device='cuda') for _ in range(3)] + [torch.empty(3 * cin, device='cuda')]
    DSC = hip.Tensor(outs[0].data_ptr(), n, h, w, cin, cin)      # the finished gradient (contiguous copy)
    bws_bytes = int(hip.lib.y3_bn_bwd_workspace(M, cin))
    bws = torch.zeros(bws_bytes, device='cuda', dtype=torch.uint8)
    hip.check(hip.lib.y3_bn_bwd_stats(DSC, A, None, 0, gd.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 0.2, ref[0].data_ptr(), ref[1].data_ptr(),
                                      ref[2].data_ptr(), ref[3].data_ptr(), bws.data_ptr(), bws_bytes, stream()))
    hip.check(hip.lib.y3_bn_bwd_finalize_tiles(part.data_ptr(), tiles, cin, M, gd.data_ptr(), mean.data_ptr(), rstd.data_ptr(), 0.2,
                                               got[0].data_ptr(), got[1].data_ptr(), got[2].data_ptr(), got[3].data_ptr(), stream()))
    for name, r, x in zip(('dgamma', 'dbeta', 'dbias', 'coef'), ref, got):
        if name == 'dbias':
            # scale of the sums it is a difference of: sum |dz| <= sum |k1 dy| + ...
            scale = float((ref[3][:cin].abs() * outs[0].abs().sum(dim=(0, 1, 2))).max())
            assert_close(x.cpu(), r.cpu(), rtol=0, atol=1e-4 * scale, what=name)
        else:
            assert_close(x.cpu(), r.cpu(), rtol=2e-5, what=name)
    # the apply kernel with the residual fan-in riding along
    _, dzr = nhwc_buf(n, h, w, cin)
    _, dzf = nhwc_buf(n, h, w, cin)
    _, drv = nhwc_buf(n, h, w, cin, ld=cin + 12, fill=0.0)
    DR = hip.Tensor(drv.data_ptr(), n, h, w, cin, cin + 12)
    hip.check(hip.lib.y3_bn_bwd_apply(DSC, A, ref[3].data_ptr(), 0.2, hip.Tensor(dzr.data_ptr(), n, h, w, cin, cin), stream()))
    for acc in (0, 1):
        hip.check(hip.lib.y3_bn_bwd_apply_fanin(DSC, A, ref[3].data_ptr(), 0.2, hip.Tensor(dzf.data_ptr(), n, h, w, cin, cin), DR, acc, stream()))
    assert torch.equal(dzr, dzf) and torch.equal(drv.cpu(), 2 * outs[0].cpu())


# the 23 distinct convolution shapes of the network at 416 x 416 (cin as the kernels see it: 3 -> 4), batch 1
APP_A = [(416, 4, 32, 3, 1), (416, 32, 64, 3, 2), (208, 64, 32, 1, 1), (208, 32, 64, 3, 1), (208, 64, 128, 3, 2), (104, 128, 64, 1, 1),
         (104, 64, 128, 3, 1), (104, 128, 256, 3, 2), (52, 256, 128, 1, 1), (52, 128, 256, 3, 1), (52, 256, 512, 3, 2), (26, 512, 256, 1, 1),
         (26, 256, 512, 3, 1), (26, 512, 1024, 3, 2), (13, 1024, 512, 1, 1), (13, 512, 1024, 3, 1), (13, 512, 512, 1, 1), (26, 1024, 256, 1, 1),
         (26, 256, 256, 1, 1), (52, 512, 128, 1, 1), (13, 1024, 14, 1, 1), (26, 512, 14, 1, 1), (52, 256, 14, 1, 1)]


@pytest.mark.parametrize('shape', APP_A)
def test_conv_x3_error_against_fp64_is_that_of_the_f32_instruction(hip, shape):
    """The reporting rule for Y3_CONV_X3 (quoted in DESIGN.md): on every Appendix-A shape the x3 kernels take, forward
    and data gradient, the maximum error against fp64 is at most 2x that of the v_mfma_f32 kernel ON THE SAME INPUTS
    (measured: 0.5-0.9x -- the pieces are exact and the 32x32x16 instruction rounds once per 16 products).  Shapes x3 does not
    take (the RGB layer, the 14-channel heads, the stride-2 data gradient into 32 channels) must be refused loudly, not computed
    some other way."""
    from util import nhwc_buf, stream
    hw, cin, cout, k, s = shape
    n = 1 if hw >= 104 else 2
    oh = -(-hw // s)
    g = torch.Generator().manual_seed(hw * 7 + cin + cout + k)
    x = torch.randn(n, cin, hw, hw, generator=g)
    wk = torch.randn(k, k, cin, cout, generator=g) * (1.0 / (k * k * cin) ** 0.5)
    dy = torch.randn(n, cout, oh, oh, generator=g)
    _, xv = nhwc_buf(n, hw, hw, cin)
    xv.copy_(x.permute(0, 2, 3, 1))
    cld = (cout + 3) // 4 * 4              # pixel pitch: a multiple of 4 floats (the 14-channel heads)
    _, dyv = nhwc_buf(n, oh, oh, cout, ld=cld)
    dyv.copy_(dy.permute(0, 2, 3, 1))
    w_keras, w_t = wk.contiguous().cuda(), wk.permute(0, 1, 3, 2).contiguous().cuda()
    from util import x3_planes
    # (planes exist only for K per row a multiple of 16: the shapes x3 refuses get the fp32 tensor, which the call must reject before reading it)
    p_keras = x3_planes(hip, w_keras) if cout % 16 == 0 else w_keras
    p_t = x3_planes(hip, w_t) if cin % 16 == 0 else w_t
    X, DY = hip.Tensor(xv.data_ptr(), n, hw, hw, cin, cin), hip.Tensor(dyv.data_ptr(), n, oh, oh, cout, cld)
    m = n * oh * oh
    ok = bool(hip.lib.y3_conv2d_x3_ok(m, cin, k * k, cout))
    assert ok == (cin % 16 == 0 and cout >= 32)
    # forward
    xr = x.double().requires_grad_(True)
    ref = _conv_ref(xr, wk, None, k, s)
    outs = {}
    for name, flag, wt in (('f32', 0, w_keras), ('x3', hip.CONV_X3, p_t)):
        _, yv = nhwc_buf(n, oh, oh, cout, ld=cld)
        wsb = int(hip.lib.y3_conv2d_fwd_workspace_x(m, cin, k, cout, flag))
        ws = torch.zeros(wsb // 4 + 4, device='cuda')
        rc = hip.lib.y3_conv2d_fwd(X, wt.data_ptr(), None, k, s, hip.Tensor(yv.data_ptr(), n, oh, oh, cout, cld), flag, 0.0, None, None, None, None,
                                   ws.data_ptr(), wsb, stream())
        if flag and not ok:
            assert rc != 0, 'x3 accepted a shape y3_conv2d_x3_ok() refuses'
            continue
        hip.check(rc, 'conv fwd ' + name)
        outs[name] = float((yv.cpu().permute(0, 3, 1, 2).double() - ref.detach()).abs().max())
    scale = float(ref.detach().abs().max())
    assert outs['f32'] <= 2e-5 * scale
    if ok:
        assert outs['x3'] <= max(2.0 * outs['f32'], 2e-7 * scale), 'forward: x3 error %.3e vs f32 %.3e' % (outs['x3'], outs['f32'])
    # data gradient (not for the RGB layer: the network never asks for it)
    if cin == 4:
        return
    ref.backward(dy.double())
    refd = xr.grad.permute(0, 2, 3, 1)
    okd = bool(hip.lib.y3_conv2d_dgrad_x3_ok(DY, k, s, hip.Tensor(0, n, hw, hw, cin, cin)))
    assert okd == (bool(hip.lib.y3_conv2d_x3_ok(n * hw * hw, cout, k * k, cin)) if s == 1 else (cin >= 64 and cout % 16 == 0))
    outs = {}
    for name, flag, wt in (('f32', 0, w_t), ('x3', hip.CONV_X3, p_keras)):
        _, dxv = nhwc_buf(n, hw, hw, cin, fill=0.0)
        DX = hip.Tensor(dxv.data_ptr(), n, hw, hw, cin, cin)
        wsb = int(hip.lib.y3_conv2d_dgrad_workspace_x(DY, k, s, DX, flag))
        ws = torch.zeros(wsb // 4 + 4, device='cuda')
        rc = hip.lib.y3_conv2d_dgrad(DY, wt.data_ptr(), k, s, DX, flag, ws.data_ptr(), wsb, stream())
        if flag and not okd:
            assert rc != 0, 'x3 data gradient accepted a shape it is not built for'
            continue
        hip.check(rc, 'conv dgrad ' + name)
        outs[name] = float((dxv.cpu().double() - refd).abs().max())
    scale = float(refd.abs().max())
    assert outs['f32'] <= 2e-5 * scale
    if okd:
        assert outs['x3'] <= max(2.0 * outs['f32'], 2e-7 * scale), 'data gradient: x3 error %.3e vs f32 %.3e' % (outs['x3'], outs['f32'])


WGRAD_CASES = CONV_CASES[:6] + CONV_CASES[9:10] + [(8, 52, 52, 128, 256, 1, 1), (2, 64, 64, 32, 64, 3, 2)]


@pytest.mark.parametrize('arith', ['f32', 'x3'])
@pytest.mark.parametrize('case', WGRAD_CASES + [(8, 26, 26, 256, 512, 3, 1), (4, 52, 52, 128, 256, 3, 1), (2, 30, 26, 64, 192, 3, 2)])
def test_conv_wgrad(hip, case, arith):
    """y3_conv2d_wgrad vs autograd; 5e-5 * max|ref| (fp32 sums over up to 21k pixels, fp32 slab combine).  arith = 'x3':
    y3_conv2d_wgrad_x with Y3_CONV_X3 (both operands split into three bf16 pieces, transposed LDS reads), same tolerance."""
    from util import nhwc_buf, stream, assert_close
    n, h, w, cin, cout, k, s = case
    x3 = 0
    if arith == 'x3':
        if not hip.lib.y3_conv2d_wgrad_x3_ok(n * (-(-h // s)) * (-(-w // s)), cin, k, cout):
            pytest.skip('the x3 kernel gradient does not take this shape')
        x3 = hip.CONV_X3
    g = torch.Generator().manual_seed(13)
    x = torch.randn(n, cin, h, w, generator=g)
    wk = (torch.randn(k, k, cin, cout, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    xx = _pad_same(x.double(), k, s)
    y = F.conv2d(xx, wk.permute(3, 2, 0, 1), None, stride=s)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    oh, ow = y.shape[2], y.shape[3]
    cld = (cout + 3) // 4 * 4
    _, sv = nhwc_buf(n, h, w, cin, ld=cin + 4)
    sv.copy_(x.permute(0, 2, 3, 1))
    _, ddv = nhwc_buf(n, oh, ow, cout, ld=cld, fill=0.0)
    ddv.copy_(dy.permute(0, 2, 3, 1))
    src = hip.Tensor(sv.data_ptr(), n, h, w, cin, cin + 4)
    dd = hip.Tensor(ddv.data_ptr(), n, oh, ow, cout, cld)
    wsb = int(hip.lib.y3_conv2d_wgrad_workspace_x(src, dd, k, s, x3))
    ws = torch.zeros(wsb // 4 + 4, device='cuda')        # tickets in the head: zeroed once (yolo3hip.h)
    dw = torch.full((k, k, cin, cout), float('nan'), device='cuda')
    hip.check(hip.lib.y3_conv2d_wgrad_x(src, dd, k, s, dw.data_ptr(), x3, ws.data_ptr(), wsb, stream()))
    assert_close(dw.cpu(), wk.grad, rtol=5e-5, what='wgrad')


@pytest.mark.parametrize('shape', [(2, 13, 13, 64), (8, 52, 52, 128), (1, 104, 104, 32), (2, 7, 9, 1024), (2, 26, 26, 512), (2, 208, 208, 64), (8, 52, 52, 256)])
def test_batchnorm_train_fwd_bwd(hip, shape):
    """stats finalize + apply + backward (stats with the in-launch finalize + fused residual fan-in, apply) vs fp64 autograd of BN(lrelu(z))."""
    from util import nhwc_buf, stream, assert_close
    n, h, w, c = shape
    g = torch.Generator().manual_seed(17)
    z = torch.randn(n, h, w, c, generator=g, dtype=torch.float64, requires_grad=True)
    gamma = (torch.rand(c, generator=g, dtype=torch.float64) + 0.5).requires_grad_(True)
    beta = torch.randn(c, generator=g, dtype=torch.float64).requires_grad_(True)
    resid = torch.randn(n, h, w, c, generator=g)
    a = F.leaky_relu(z, 0.2)
    mean = a.mean(dim=(0, 1, 2))
    var = a.var(dim=(0, 1, 2), unbiased=False)
    y = (a - mean) * torch.rsqrt(var + 1e-3) * gamma + beta + resid.double()
    dy = torch.randn(n, h, w, c, generator=g)
    y.backward(dy.double())
    M = n * h * w
    # forward: feed the conv epilogue's partial sums (here computed in torch) then finalize + apply
    ad = a.detach().float().cuda().contiguous()
    tiles = 7
    chunks = torch.chunk(ad.view(M, c), tiles, dim=0)
    tiles = len(chunks)
    stats = torch.stack([torch.stack([ch.sum(0), (ch * ch).sum(0)]) for ch in chunks]).contiguous()
    gd, bd = gamma.detach().float().cuda(), beta.detach().float().cuda()
    mm, mv = torch.zeros(c, device='cuda'), torch.ones(c, device='cuda')
    smean, srstd, scale, shift = (torch.empty(c, device='cuda') for _ in range(4))
    hip.check(hip.lib.y3_bn_stats_finalize(stats.data_ptr(), tiles, c, M, gd.data_ptr(), bd.data_ptr(), 1e-3, 0.99, mm.data_ptr(), mv.data_ptr(),
                                           smean.data_ptr(), srstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), stream()))
    assert_close(smean.cpu(), mean.detach(), rtol=1e-5, atol=1e-6, what='batch mean')
    assert_close(srstd.cpu(), torch.rsqrt(var.detach() + 1e-3), rtol=1e-5, what='rstd')
    assert_close(mm.cpu(), 0.01 * mean.detach(), rtol=1e-5, atol=1e-7, what='moving mean')
    assert_close(mv.cpu(), 0.99 + 0.01 * var.detach() * M / (M - 1), rtol=1e-5, what='moving var')
    _, rv = nhwc_buf(n, h, w, c)
    rv.copy_(resid)
    _, yv = nhwc_buf(n, h, w, c, ld=c + 8)
    A = hip.Tensor(ad.data_ptr(), n, h, w, c, c)
    hip.check(hip.lib.y3_bn_apply(A, scale.data_ptr(), shift.data_ptr(), hip.Tensor(rv.data_ptr(), n, h, w, c, c),
                                  hip.Tensor(yv.data_ptr(), n, h, w, c, c + 8), stream()))
    assert_close(yv.cpu(), y.detach(), rtol=1e-5, what='bn apply')
    # backward
    _, dyv = nhwc_buf(n, h, w, c, ld=c + 4)
    dyv.copy_(dy)
    DY = hip.Tensor(dyv.data_ptr(), n, h, w, c, c + 4)
    ws_bytes = int(hip.lib.y3_bn_bwd_workspace(M, c))
    assert ws_bytes > 1024
    pws = torch.zeros(ws_bytes, device='cuda', dtype=torch.uint8)       # tickets zero before the first launch
    dg, db, dbias, coef = torch.empty(c, device='cuda'), torch.empty(c, device='cuda'), torch.empty(c, device='cuda'), torch.empty(3 * c, device='cuda')
    # the residual fan-in rides along: dres = dy on the first launch, dres += dy on the second (which also proves the
    # tickets were left at zero and that the sums do not depend on which workgroup finishes last)
    _, drv = nhwc_buf(n, h, w, c, ld=c + 12)
    DR = hip.Tensor(drv.data_ptr(), n, h, w, c, c + 12)
    outs = []
    for acc in (0, 1):
        hip.check(hip.lib.y3_bn_bwd_stats(DY, A, DR, acc, gd.data_ptr(), smean.data_ptr(), srstd.data_ptr(), 0.2, dg.data_ptr(), db.data_ptr(),
                                          dbias.data_ptr(), coef.data_ptr(), pws.data_ptr(), ws_bytes, stream()))
        outs.append(torch.cat([dg, db, dbias, coef]).cpu())
    assert torch.equal(outs[0], outs[1]), 'bn_bwd_stats is not reproducible'
    assert torch.equal(drv.cpu(), 2 * dyv.cpu()), 'fused residual fan-in'
    assert int(pws[:1024].view(torch.int32).abs().sum()) == 0, 'tickets not reset'
    hip.check(hip.lib.y3_bn_bwd_stats(DY, A, None, 0, gd.data_ptr(), smean.data_ptr(), srstd.data_ptr(), 0.2, dg.data_ptr(), db.data_ptr(),
                                      dbias.data_ptr(), coef.data_ptr(), pws.data_ptr(), ws_bytes, stream()))
    assert torch.equal(torch.cat([dg, db, dbias, coef]).cpu(), outs[0])
    _, dzv = nhwc_buf(n, h, w, c)
    hip.check(hip.lib.y3_bn_bwd_apply(DY, A, coef.data_ptr(), 0.2, hip.Tensor(dzv.data_ptr(), n, h, w, c, c), stream()))
    assert_close(dg.cpu(), gamma.grad, rtol=1e-4, what='dgamma')
    assert_close(db.cpu(), beta.grad, rtol=1e-4, what='dbeta')
    assert_close(dzv.cpu(), z.grad, rtol=1e-4, what='dz')
    assert_close(dbias.cpu(), z.grad.sum(dim=(0, 1, 2)), rtol=1e-4, atol=1e-4 * float(z.grad.abs().sum(dim=(0, 1, 2)).max()), what='dbias')


def test_bn_fold_inference(hip):
    from util import stream, assert_close
    c = 96
    g = torch.Generator().manual_seed(3)
    ga, be, mu = (torch.randn(c, generator=g) for _ in range(3))
    va = torch.rand(c, generator=g) + 0.1
    d = [t.cuda() for t in (ga, be, mu, va)]
    sc, sh = torch.empty(c, device='cuda'), torch.empty(c, device='cuda')
    hip.check(hip.lib.y3_bn_fold_inference(d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), 1e-3, c, sc.data_ptr(), sh.data_ptr(), stream()))
    rs = ga.double() / torch.sqrt(va.double() + 1e-3)
    assert_close(sc.cpu(), rs, rtol=1e-6)
    assert_close(sh.cpu(), be.double() - mu.double() * rs, rtol=1e-6)


def test_upsample_sum2x(hip):
    """All-ones Conv2DTranspose (Q3) forward / backward vs the literal conv_transpose2d."""
    from util import nhwc_buf, stream, assert_close
    n, h, w, c = 2, 5, 7, 64
    g = torch.Generator().manual_seed(19)
    x = torch.randn(n, c, h, w, generator=g, dtype=torch.float64, requires_grad=True)
    y = F.conv_transpose2d(x, torch.ones(c, c, 2, 2, dtype=torch.float64), stride=2)
    dy = torch.randn(y.shape, generator=g)
    y.backward(dy.double())
    _, xv = nhwc_buf(n, h, w, c)
    xv.copy_(x.detach().permute(0, 2, 3, 1))
    _, ov = nhwc_buf(n, 2 * h, 2 * w, c, ld=2 * c)        # written into the first half of a concat buffer
    hip.check(hip.lib.y3_upsample_sum2x_fwd(hip.Tensor(xv.data_ptr(), n, h, w, c, c), hip.Tensor(ov.data_ptr(), n, 2 * h, 2 * w, c, 2 * c), stream()))
    assert_close(ov.cpu().permute(0, 3, 1, 2), y.detach(), rtol=1e-5, what='upsample fwd')
    _, dov = nhwc_buf(n, 2 * h, 2 * w, c, ld=2 * c)
    dov.copy_(dy.permute(0, 2, 3, 1))
    _, dxv = nhwc_buf(n, h, w, c)
    hip.check(hip.lib.y3_upsample_sum2x_bwd(hip.Tensor(dov.data_ptr(), n, 2 * h, 2 * w, c, 2 * c), hip.Tensor(dxv.data_ptr(), n, h, w, c, c), stream()))
    assert_close(dxv.cpu().permute(0, 3, 1, 2), x.grad, rtol=1e-5, what='upsample bwd')


def test_data_movement(hip):
    from util import nhwc_buf, stream
    n, h, w, c = 2, 6, 5, 3
    g = torch.Generator().manual_seed(23)
    x = torch.randn(n, c, h, w, generator=g).cuda()
    _, dv = nhwc_buf(n, h, w, 4)
    hip.check(hip.lib.y3_nchw_to_nhwc(x.data_ptr(), n, c, h, w, hip.Tensor(dv.data_ptr(), n, h, w, 4, 4), stream()))
    assert torch.equal(dv[..., :3], x.permute(0, 2, 3, 1)) and (dv[..., 3] == 0).all()
    back = torch.empty(n, 4, h, w, device='cuda')
    hip.check(hip.lib.y3_nhwc_to_nchw(hip.Tensor(dv.data_ptr(), n, h, w, 4, 4), back.data_ptr(), stream()))
    assert torch.equal(back[:, :3], x)
    _, a = nhwc_buf(n, h, w, 8, ld=16, fill=1.0)
    _, b = nhwc_buf(n, h, w, 8, ld=8, fill=2.0)
    A, B = hip.Tensor(a.data_ptr(), n, h, w, 8, 16), hip.Tensor(b.data_ptr(), n, h, w, 8, 8)
    hip.check(hip.lib.y3_add_inplace(A, B, stream()))
    assert (b == 3.0).all()
    hip.check(hip.lib.y3_copy(B, A, stream()))
    assert (a == 3.0).all()
    f = torch.empty(1001, device='cuda')
    hip.check(hip.lib.y3_fill(f.data_ptr(), 1001, 2.5, stream()))
    assert (f == 2.5).all()
    _, m = nhwc_buf(2, 13, 13, 14, ld=16, fill=0.0)
    vals = torch.randn(2, 13, 13, 14, generator=g)
    m.copy_(vals)
    out = torch.empty(14, device='cuda')
    hip.check(hip.lib.y3_colsum(hip.Tensor(m.data_ptr(), 2, 13, 13, 14, 16), out.data_ptr(), stream()))
    np.testing.assert_allclose(out.cpu().numpy(), vals.double().sum(dim=(0, 1, 2)).numpy(), rtol=1e-5, atol=1e-5)


def test_errors_are_reported(hip):
    """Bad arguments return an error code and a message; nothing is launched."""
    t = hip.Tensor(0, 1, 1, 1, 4, 4)
    rc = hip.lib.y3_conv2d_fwd(t, None, None, 3, 1, t, 0, 0.0, None, None, None, None, None, 0, None)
    assert rc == -1 and b'null' in hip.lib.y3_last_error()
    with pytest.raises(hip.HipError):
        hip.check(rc, 'y3_conv2d_fwd')


def test_dirty_ticket_header_is_detected(tmp_path):
    """a conv workspace whose ticket header was never zeroed fails silently (stale output, Y3_OK).  With
    Y3_CHECK_TICKETS=1 the launch reports it.  In a process of its own: the switch is read once per process."""
    import subprocess
    import sys
    code = r'''
import sys, torch
sys.path.insert(0, %r)
from yolo3 import _hip
lib = _hip.lib
n, h, w, cin, cout, k = 2, 13, 13, 512, 1024, 3
x = torch.randn(n, h, w, cin, device="cuda")
wt = torch.randn(k * k * cin * cout, device="cuda") * 0.02
b = torch.zeros(cout, device="cuda")
y = torch.empty(n, h, w, cout, device="cuda")
wsb = int(lib.y3_conv2d_fwd_workspace(n * h * w, cin, k, cout))
assert wsb > 0
st = torch.cuda.current_stream().cuda_stream
src, dst = _hip.Tensor(x.data_ptr(), n, h, w, cin, cin), _hip.Tensor(y.data_ptr(), n, h, w, cout, cout)
run = lambda ws: lib.y3_conv2d_fwd(src, wt.data_ptr(), b.data_ptr(), k, 1, dst, 1, 0.2, None, None, None, None, ws.data_ptr(), wsb, st)
good = torch.zeros(wsb // 4 + 4, device="cuda")
assert run(good) == 0 and run(good) == 0                      # a zeroed header stays zero: launch after launch
bad = torch.zeros(wsb // 4 + 4, device="cuda")
bad.view(torch.int32)[7] = 3                                  # one stale ticket
rc = run(bad)
msg = lib.y3_last_error().decode()
assert rc == -1 and "ticket 7" in msg and "is 3" in msg, (rc, msg)
print("detected:", msg[:60])
''' % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'object-detection-yolov3_amd')
    env = dict(os.environ, Y3_CHECK_TICKETS='1')
    env.pop('Y3_NO_FAST', None)            # the generic kernel has no tickets: nothing to detect there
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'detected:' in r.stdout, r.stdout[-1000:] + r.stderr[-2000:]


def test_decode_matches_oracle(hip):
    """y3_decode_fwd vs oracle.model.decode (fp64); 1e-5 relative to the box scale."""
    from oracle import model as om
    from util import nhwc_buf, stream, assert_close
    n, A, K, H, W = 3, 2, 2, 416, 416
    anchors = [(64, 384), (384, 64)]
    g = torch.Generator().manual_seed(29)
    fms = [torch.randn(n, A * (5 + K), H // s, W // s, generator=g) * 1.5 for s in (32, 16, 8)]
    ref = om.decode([f.double() for f in fms], (H, W, 3), anchors, K)
    views = []
    arr = (hip.Tensor * 3)()
    for i, f in enumerate(fms):
        _, v = nhwc_buf(n, f.shape[2], f.shape[3], 14, ld=16)
        v.copy_(f.permute(0, 2, 3, 1))
        views.append(v)
        arr[i] = hip.Tensor(v.data_ptr(), n, f.shape[2], f.shape[3], 14, 16)
    nb = ref.shape[1]
    out = torch.empty(n, nb, 5 + K, device='cuda')
    hip.check(hip.lib.y3_decode_fwd(arr, 3, hip.float_array([v for a in anchors for v in a]), A, K, H, W, out.data_ptr(), stream()))
    assert nb == 7098
    assert_close(out.cpu()[..., :4], ref[..., :4], rtol=1e-5, what='boxes')
    assert_close(out.cpu()[..., 4:], ref[..., 4:], rtol=1e-5, atol=2e-6, what='scores')


def _labels(rng, n, img, anchors, K, per_image=3):
    from yolo3.imagereader import format_boxes
    labs = [[], [], []]
    for _ in range(n):
        k = rng.integers(0, per_image + 1)
        wh = rng.integers(20, img // 2, (k, 2))
        xy = np.stack([rng.integers(0, img - wh[:, 0]), rng.integers(0, img - wh[:, 1])], 1) if k else np.zeros((0, 2), int)
        boxes = np.concatenate([xy, wh, rng.integers(0, K, (k, 1))], 1).astype(np.int32)
        lab = format_boxes(boxes, (img, img, 3), anchors, K)
        for i in range(3):
            labs[i].append(lab[i])
    return [np.stack(l) for l in labs]


@pytest.mark.parametrize('empty', [False, True])
def test_loss_fwd_bwd_matches_oracle(hip, empty):
    """y3_loss_fwd_bwd vs oracle.model.loss_layer + autograd in fp64; losses 1e-5, gradients 1e-4 relative."""
    from oracle import model as om
    from util import nhwc_buf, stream, assert_close
    n, A, K, img = 4, 2, 2, 416
    anchors = [(64, 384), (384, 64)]
    rng = np.random.default_rng(31)
    gts = _labels(rng, n, img, anchors, K, per_image=0 if empty else 4)
    g = torch.Generator().manual_seed(37)
    loss4 = torch.zeros(4, device='cuda')
    ws = torch.zeros(int(hip.lib.y3_loss_workspace_bytes()) // 4 + 4, device='cuda')
    ref_parts = np.zeros(4)
    gbs = 16.0
    for si, s in enumerate((32, 16, 8)):
        G = img // s
        fm = (torch.randn(n, A * (5 + K), G, G, generator=g) * 1.2).double().requires_grad_(True)
        gt = torch.from_numpy(gts[si])
        # put one prediction near the origin so the ignore mask (Q7) actually fires somewhere
        parts = om.loss_layer(fm, gt.double(), (img, img, 3), anchors, K)
        total = sum(parts) / gbs
        total.backward()
        ref_parts += np.array([float(p) for p in parts])
        _, fv = nhwc_buf(n, G, G, 14, ld=16, fill=0.0)
        fv.copy_(fm.detach().float().permute(0, 2, 3, 1))
        _, dv = nhwc_buf(n, G, G, 14, ld=16, fill=0.0)
        gd = gt.float().cuda().contiguous()
        hip.check(hip.lib.y3_loss_fwd_bwd(hip.Tensor(fv.data_ptr(), n, G, G, 14, 16), gd.data_ptr(), hip.float_array([v for a in anchors for v in a]),
                                          A, K, img, img, gbs, loss4.data_ptr(), hip.Tensor(dv.data_ptr(), n, G, G, 14, 16), ws.data_ptr(), stream()))
        assert_close(dv.cpu().permute(0, 3, 1, 2), fm.grad, rtol=1e-4, what='dfm scale %d' % s)
    assert_close(loss4.cpu(), ref_parts, rtol=2e-5, what='loss parts (xy, wh, obj, class)')


@pytest.mark.parametrize('tag', ['sq', 'rect'])
def test_decode_and_loss_match_reference_text_fixtures(hip, golden_dir, tag):
    """y3_decode_fwd and y3_loss_fwd_bwd against tests/golden/model_fwd.npz: the reference's own reorg_layer /
    convert_feature_map_to_inference_detections / loss_layer text (model.py:122-354) executed over NumPy float32 stand-ins of
    the tf.* calls (make_golden_model.py; TensorFlow itself is absent, so this pins the transcription, not TF arithmetic).
    Square 416 input with 2 anchors and a 96 x 160 input with 3 anchors / 3 classes (stride quirk Q6 visible), V > 0 and
    V = 0.  Rows 1e-5 of the box scale, losses 1e-4."""
    from util import nhwc_buf, stream, assert_close
    z = np.load(os.path.join(golden_dir, 'model_fwd.npz'))
    H, W, C, K, n = (int(v) for v in z[tag + '_meta'])
    anchors = [tuple(float(v) for v in a) for a in z[tag + '_anchors']]
    A = len(anchors)
    D = A * (5 + K)
    ld = (D + 3) // 4 * 4
    anc = hip.float_array([v for a in anchors for v in a])
    arr = (hip.Tensor * 3)()
    views = []
    for i in range(3):
        f = torch.from_numpy(z['%s_fm%d' % (tag, i)])
        _, v = nhwc_buf(n, f.shape[2], f.shape[3], D, ld=ld, fill=0.0)
        v.copy_(f.permute(0, 2, 3, 1))
        views.append(v)
        arr[i] = hip.Tensor(v.data_ptr(), n, f.shape[2], f.shape[3], D, ld)
    ref = z[tag + '_rows']
    out = torch.empty(n, ref.shape[1], 5 + K, device='cuda')
    hip.check(hip.lib.y3_decode_fwd(arr, 3, anc, A, K, H, W, out.data_ptr(), stream()))
    assert_close(out.cpu()[..., :4], ref[..., :4], rtol=1e-5, what='boxes')
    assert_close(out.cpu()[..., 4:], ref[..., 4:], rtol=1e-5, atol=2e-6, what='scores')
    ws = torch.zeros(int(hip.lib.y3_loss_workspace_bytes()) // 4 + 4, device='cuda')
    for suffix in ('', '_v0'):
        loss4 = torch.zeros(4, device='cuda')
        want = np.zeros(4)
        for i in range(3):
            gt = torch.from_numpy(z['%s_gt%d' % (tag, i)])
            if suffix:
                gt = torch.zeros_like(gt)
            want += z['%s_loss%d%s' % (tag, i, suffix)]
            gh, gw = gt.shape[1], gt.shape[2]
            _, dv = nhwc_buf(n, gh, gw, D, ld=ld, fill=0.0)
            gd = gt.float().cuda().contiguous()
            hip.check(hip.lib.y3_loss_fwd_bwd(hip.Tensor(views[i].data_ptr(), n, gh, gw, D, ld), gd.data_ptr(), anc, A, K, H, W, float(n),
                                              loss4.data_ptr(), hip.Tensor(dv.data_ptr(), n, gh, gw, D, ld), ws.data_ptr(), stream()))
            assert bool(torch.isfinite(dv).all())
        assert_close(loss4.cpu(), want, rtol=1e-4, what='loss parts (xy, wh, obj, class)%s' % suffix)
        if not suffix:
            assert_close(float(loss4.sum()), float(z[tag + '_compute_loss'][0]), rtol=1e-4, what='compute_loss total')


def test_adam_matches_oracle(hip):
    """y3_adam_step vs oracle.model.AdamState (Keras form, eps outside the bias correction); 3 steps, 1e-6 relative."""
    from oracle import model as om
    from util import stream, assert_close
    count = 100003
    g = torch.Generator().manual_seed(41)
    p0 = torch.randn(count, generator=g)
    ref_p = p0.clone().double()
    st = om.AdamState([ref_p], 1e-3)
    p, m, v = p0.cuda(), torch.zeros(count, device='cuda'), torch.zeros(count, device='cuda')
    lr = torch.zeros(1, device='cuda')
    for _ in range(3):
        gr = torch.randn(count, generator=g) * 0.01
        st.step([ref_p], [gr.double()])
        lr.fill_(st.lr_t())
        gd = gr.cuda()
        hip.check(hip.lib.y3_adam_step(p.data_ptr(), gd.data_ptr(), m.data_ptr(), v.data_ptr(), count, lr.data_ptr(), 0.9, 0.999, 1e-7, stream()))
    assert_close(p.cpu(), ref_p, rtol=1e-6, what='adam params')
    assert_close(m.cpu(), st.m[0], rtol=1e-5, what='adam m')
    # 1 - float32(0.999) differs from 0.001 by 1.3e-5 relative: inherent to fp32 hyper-parameters (TF's kernel does the same)
    assert_close(v.cpu(), st.v[0], rtol=5e-5, what='adam v')


def _golden_nms(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, 'nms_*.npz')))


def test_filter_small_boxes_matches_reference(hip, golden_dir):
    """bbox_utils.filter_small_boxes on the GPU keeps exactly the rows the reference keeps (strict '>', row order):
    checked through the goldens' filtered row sets (the reference's own filter output feeds its NMS there)."""
    from yolo3 import bbox_utils
    from oracle import nms as onms
    for path in _golden_nms(golden_dir):
        z = np.load(path)
        if 'rows' not in z.files:
            continue
        rows, mb = z['rows'], float(z['min_box'])
        got = bbox_utils.filter_small_boxes(rows, mb)
        want = onms.filter_small_boxes(rows, mb)          # pinned to the reference by tests/test_cpu_oracle.py
        assert got.dtype == rows.dtype and np.array_equal(got, want), path
    edge = np.asarray([[0, 0, 32, 40, 1], [0, 0, 32.0001, 32.0001, 2], [5, 5, 100, 37, 3], [5, 5, 38, 38, 4]], np.float32)
    assert bbox_utils.filter_small_boxes(edge, 32)[:, 4].tolist() == [2.0, 4.0]
    assert bbox_utils.filter_small_boxes(np.zeros((0, 7), np.float32), 32).shape == (0, 7)


def test_compute_iou_matches_reference(hip, golden_dir):
    """bbox_utils.compute_iou on the GPU == the reference's 64 x 64 IoU matrix (nms_units.npz), bit for bit incl. the
    0/0 -> NaN entries of zero-area boxes and the IoU-exactly-at-threshold pair."""
    from yolo3 import bbox_utils
    u = np.load(os.path.join(golden_dir, 'nms_units.npz'))
    boxes = u['boxes'].astype(np.float32)
    got = np.stack([bbox_utils.compute_iou(boxes[i], boxes) for i in range(64)])
    want = u['ious'].astype(np.float32)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    assert np.array_equal(np.nan_to_num(got, nan=-1).view(np.uint32), np.nan_to_num(want, nan=-1).view(np.uint32))
    assert bbox_utils.compute_iou(boxes[0], np.zeros((0, 4), np.float32)).shape == (0,)


def test_nms_matches_reference_goldens(hip, golden_dir):
    """Class-wise NMS kernel vs the outputs of the reference's bbox_utils (tests/golden): keep indices,
    labels and scores must be IDENTICAL (integer / bit-exact)."""
    from yolo3 import bbox_utils
    names = [p for p in _golden_nms(golden_dir) if 'units' not in p]
    assert len(names) >= 8
    for path in names:
        z = np.load(path)
        rows = torch.from_numpy(z['rows']).cuda()[None]
        b, s, l, keep = bbox_utils.detect(rows, float(z['min_box']))[0]
        if 'keep' not in z:
            assert b is None, path
            continue
        assert np.array_equal(keep, z['keep']), path
        assert np.array_equal(l, z['labels']), path
        assert np.array_equal(s.view(np.uint32), z['scores'].view(np.uint32)), path
        assert np.array_equal(b, z['boxes']), path


def test_nms_batched_and_api(hip, golden_dir):
    """Several images in one launch + the reference-signature wrappers."""
    from yolo3 import bbox_utils
    from oracle import nms as onms
    za = np.load(os.path.join(golden_dir, 'nms_sparse416_k2.npz'))
    zb = np.load(os.path.join(golden_dir, 'nms_dense416_k2.npz'))
    rows = torch.from_numpy(np.stack([za['rows'], zb['rows'], za['rows']])).cuda()
    res = bbox_utils.detect(rows, 32.0)
    assert np.array_equal(res[0][3], za['keep']) and np.array_equal(res[1][3], zb['keep']) and np.array_equal(res[2][3], za['keep'])
    f = onms.filter_small_boxes(za['rows'], 32)
    b, s, l = bbox_utils.per_class_nms(f[:, 0:4], f[:, 4:5], f[:, 5:])
    assert np.array_equal(b, za['boxes']) and np.array_equal(l, za['labels'])
    u = np.load(os.path.join(golden_dir, 'nms_units.npz'))
    for thr in (0.3, 0.5, 0.0):
        assert bbox_utils.single_class_nms(u['boxes'], u['scores'], thr) == list(u['keep_%g' % thr]), thr


def test_nms_clip_and_large(hip):
    """clip-to-image option (inference.py:62-65 intent) and the > 16384-candidate global-memory path vs the oracle."""
    from yolo3 import bbox_utils
    from oracle import nms as onms
    rng = np.random.default_rng(43)
    nb = 20000
    rows = np.zeros((nb, 6), np.float32)
    c = rng.uniform(0, 800, (nb, 2))
    wh = rng.uniform(20, 90, (nb, 2))
    rows[:, 0:2] = c - wh / 2
    rows[:, 2:4] = c + wh / 2
    rows[:, 4] = rng.uniform(0.3, 1, nb)
    rows[:, 5] = rng.uniform(0.3, 1, nb)
    sc = np.sqrt(rows[:, 5] * rows[:, 4])
    _, first = np.unique(sc, return_index=True)
    rows = rows[np.sort(first)]            # drop score ties (argsort order unspecified in the reference)
    clipped = rows.copy()
    clipped[:, [0, 2]] = np.clip(clipped[:, [0, 2]], 0, 700)
    clipped[:, [1, 3]] = np.clip(clipped[:, [1, 3]], 0, 750)
    want = onms.detect_rows(clipped, 32)[0]
    got = bbox_utils.detect(torch.from_numpy(rows).cuda()[None], 32.0, clip_wh=(700, 750))[0]
    assert rows.shape[0] > 16384
    assert np.array_equal(got[3], want)


def test_zscore_matches_reference_goldens(hip, golden_dir):
    """y3_zscore vs imagereader.zscore_normalize outputs; 2e-6 absolute on O(1) values."""
    from yolo3 import imagereader
    z = np.load(os.path.join(golden_dir, 'zscore.npz'))
    for k in 'abc':
        got = imagereader.zscore_normalize(z[k])
        assert got.dtype == np.float32 and got.shape == z[k].shape
        np.testing.assert_allclose(got, z[k + '_out'], rtol=2e-6, atol=2e-6)


def test_rccl_comm_single_rank(hip):
    """y3_comm_* (RCCL behind the C ABI): a one-rank communicator -- the most a single-GPU box can host, RCCL allows one
    rank per device -- sums a gradient-sized buffer with itself (identity), asynchronously on the caller's stream; bad
    arguments are reported, not crashed on.  The multi-rank exchange is covered by the gloo world-size-2 tests of
    yolo3.parallel and by bench.py --gpus N on a multi-GPU node."""
    import ctypes as C
    from util import stream
    uid = (C.c_char * 128)()
    hip.check(hip.lib.y3_comm_unique_id(uid))
    comm = C.c_void_p()
    hip.check(hip.lib.y3_comm_init(uid, 1, 0, C.byref(comm)))
    assert comm.value
    g = torch.Generator().manual_seed(1)
    buf = torch.randn(1 << 20, generator=g).cuda()
    want = buf.clone()
    hip.check(hip.lib.y3_allreduce_sum_f32(comm, buf.data_ptr(), buf.numel(), stream()))
    hip.check(hip.lib.y3_allreduce_sum_f32(comm, buf.data_ptr(), 0, stream()))
    torch.cuda.synchronize()
    assert torch.equal(buf, want)
    hip.check(hip.lib.y3_comm_destroy(comm))
    with pytest.raises(hip.HipError):
        hip.check(hip.lib.y3_comm_init(uid, 2, 5, C.byref(comm)))
