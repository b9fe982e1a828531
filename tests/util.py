"""Helpers shared by the GPU parity tests."""
import numpy as np
import torch


def dev():
    return torch.device('cuda:0')


def stream():
    return torch.cuda.current_stream().cuda_stream


def nhwc_buf(n, h, w, c, ld=None, fill=None, off=0):
    """Flat CUDA buffer holding an NHWC tensor with pitch ld, plus its strided torch view."""
    ld = c if ld is None else ld
    buf = torch.full((n * h * w * ld + off,), float('nan') if fill is None else fill, dtype=torch.float32, device=dev())
    tv = torch.as_strided(buf, (n, h, w, c), (h * w * ld, w * ld, ld, 1), off)
    return buf, tv


def to_nhwc(buf_view, x_nchw):
    buf_view.copy_(x_nchw.permute(0, 2, 3, 1).to(buf_view.device))


def assert_close(got, want, rtol=1e-4, atol=None, what=''):
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    scale = np.abs(want).max() if want.size else 1.0
    atol = rtol * max(scale, 1e-30) if atol is None else atol
    err = np.abs(got - want)
    assert np.isfinite(got).all(), '%s: non-finite output' % what
    assert err.max() <= atol, '%s: max abs err %.3e > %.3e (scale %.3e)' % (what, err.max(), atol, scale)


def x3_planes(hip, w):
    """The weight operand of a Y3_CONV_X3 launch: the bf16 piece planes (y3_x3_split_weights) of a CUDA fp32 kernel [kh, kw, rows, K]
    that already has the layout the entry point wants (K contiguous per row)."""
    w = w.contiguous()
    kh, kw, rows, kpr = w.shape
    planes = torch.empty(3 * w.numel(), dtype=torch.bfloat16, device=w.device)
    hip.check(hip.lib.y3_x3_split_weights(w.data_ptr(), planes.data_ptr(), kh * kw, rows, kpr, stream()), 'y3_x3_split_weights')
    return planes
