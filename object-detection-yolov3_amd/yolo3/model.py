"""YoloV3 on MI355X: host-side mirror of the reference's ``model.YoloV3``
(/root/reference/model.py:19-540) driving hand-written HIP kernels through the
C ABI in include/yolo3hip.h.

Same constructor, methods and tensor contracts as the reference class
(model.py:423-540): NCHW z-scored images in, ``[N, Nb, 5+K]`` detections out,
three ``[N, G, G, A, 5+K]`` label tensors for the loss.  Internally everything
is NHWC fp32 in HBM (channel-contiguous = GEMM-K-contiguous), parameters /
gradients / Adam moments live in flat arenas in Keras ``trainable_weights``
order, and each step is a static list of kernel launches (replayable as one
HIP graph).  torch is used for device memory, streams and the RCCL all-reduce
only; there is no torch.nn / autograd / CPU fallback on this path.
"""
import math
import os

import numpy as np
import torch

from . import _hip
from . import streams
from ._hip import lib, check, view, EPI_LRELU, EPI_ACCUM, CONV_X3, BF16_NO_PATCH

BN_EPS = 1e-3          # Keras BatchNormalization defaults
BN_MOMENTUM = 0.99
BN_EPILOGUE_STATS = os.environ.get('Y3_BN_EPI', '1') != '0'   # BatchNorm-backward statistics from the epilogue of the data gradient that completes dy
BN_EPILOGUE_WIDE = os.environ.get('Y3_BN_EPI_WIDE', '1') != '0'   # ... also when the consumer is a stride-2 convolution or reads a concat slice
LRELU_ALPHA = 0.2      # tf.nn.leaky_relu default (App. C3)
BF16_PATCH_MIN_BYTES = 300e6   # bf16 path: a 3x3 layer that moves less (input + residual + output) stays off the patch kernels
ALIGN = 64             # arena alignment in floats (256 B)


def _round_up(v, a):
    return (v + a - 1) // a * a


class LayerSpec:
    __slots__ = ('cin', 'cin_pad', 'cout', 'k', 's', 'bn', 'w_off', 'b_off', 'g_off', 'be_off', 'end_off', 'bn_idx', 'ch_off', 'mv_off')


def build_layer_specs(in_channels, num_anchors, num_classes):
    """Conv layers in Keras creation order (model.py:356-421); see the walk in
    ``_Plan._build`` which consumes them in the same order."""
    L = []

    def conv(cin, cout, k, s=1, bn=True):
        sp = LayerSpec()
        sp.cin, sp.cout, sp.k, sp.s, sp.bn = cin, cout, k, s, bn
        sp.cin_pad = _round_up(cin, 4)
        L.append(sp)
        return cout

    def feature_block(c, reps):
        for _ in range(reps):
            conv(c, c // 2, 1)
            conv(c // 2, c, 3)
        return c

    FC = YoloV3.FILTER_COUNT
    c = conv(in_channels, FC // 32, 3)
    c = conv(c, FC // 16, 3, 2)
    c = feature_block(c, 1)
    c = conv(c, FC // 8, 3, 2)
    c = feature_block(c, 2)
    c = conv(c, FC // 4, 3, 2)
    c = feature_block(c, YoloV3.BLOCK_COUNT)
    c = conv(c, FC // 2, 3, 2)
    c = feature_block(c, YoloV3.BLOCK_COUNT)
    c = conv(c, FC, 3, 2)
    c = feature_block(c, YoloV3.BLOCK_COUNT // 2)
    D = num_anchors * (5 + num_classes)

    def yolo_block(cin, fc):
        for i in range(3):
            conv(cin if i == 0 else fc, fc // 2, 1)
            conv(fc // 2, fc, 3)

    yolo_block(FC, FC)
    conv(FC, D, 1, 1, bn=False)
    conv(FC // 2, FC // 2, 1)
    yolo_block(FC, FC // 2)
    conv(FC // 2, D, 1, 1, bn=False)
    conv(FC // 4, FC // 4, 1)
    yolo_block(FC // 2, FC // 4)
    conv(FC // 4, D, 1, 1, bn=False)
    # arena offsets (floats)
    off = 0
    bn_idx = 0
    ch = 0
    mv = 0
    for sp in L:
        sp.w_off = off
        off = _round_up(off + sp.k * sp.k * sp.cin_pad * sp.cout, ALIGN)
        sp.b_off = off
        off = _round_up(off + sp.cout, ALIGN)
        if sp.bn:
            sp.g_off = off
            off = _round_up(off + sp.cout, ALIGN)
            sp.be_off = off
            off = _round_up(off + sp.cout, ALIGN)
            sp.bn_idx = bn_idx
            bn_idx += 1
            sp.ch_off = ch                  # per-layer block [scale|shift|save_mean|save_rstd|k1|k2|k3], each cout floats
            ch = _round_up(ch + 7 * sp.cout, ALIGN)
            sp.mv_off = mv                  # offset into the moving mean / variance arrays
            mv = _round_up(mv + sp.cout, ALIGN)
        else:
            sp.g_off = sp.be_off = -1
            sp.bn_idx = -1
            sp.ch_off = sp.mv_off = -1
        sp.end_off = off
    return L, off, ch, mv


class _T:
    """An NHWC activation (or a channel slice of one) plus its gradient twin."""

    def __init__(self, buf, n, h, w, c, ld=None, off=0):
        self.buf, self.n, self.h, self.w, self.c = buf, n, h, w, c
        self.ld = c if ld is None else ld
        self.off = off
        self.v = view(buf, n, h, w, c, self.ld, off)
        self.grad = None
        self.gw = False           # gradient already holds a contribution (next writer accumulates)
        self.children = []
        self.parent = None

    @property
    def m(self):
        return self.n * self.h * self.w

    def slice(self, c0, c):
        t = _T(self.buf, self.n, self.h, self.w, c, self.ld, self.off + c0)
        t.parent = self
        self.children.append(t)
        return t

    def mark_written(self):
        self.gw = True
        for ch in self.children:
            ch.gw = True

    def torch_view(self):
        """[N,H,W,C] strided torch view (debug / export)."""
        flat = self.buf.view(-1)
        return torch.as_strided(flat, (self.n, self.h, self.w, self.c), (self.h * self.w * self.ld, self.w * self.ld, self.ld, 1), self.off)


class Mean:
    """Minimal stand-in for tf.keras.metrics.Mean (train.py:80-90)."""

    def __init__(self, name='mean', dtype=None):
        self.name = name
        self.reset_states()

    def update_state(self, value):
        self.total = self.total + (value.detach() if torch.is_tensor(value) else float(value))
        self.count += 1

    def result(self):
        if self.count == 0:
            return 0.0
        t = self.total / self.count
        return float(t.item()) if torch.is_tensor(t) else float(t)

    def reset_states(self):
        self.total = 0.0
        self.count = 0


class _Plan:
    """Static launch list for one (batch size, mode)."""

    def __init__(self, model, n, training, bf16=False):
        self.model = model
        self.n = n
        self.training = training
        self.bf16 = bool(bf16) and not training     # reduced-precision conv path: inference only (BASELINE config 5)
        self.fwd = []      # [(cfunc, args)]   stream appended at run time
        self.bwd = []
        self.keep = []     # ctypes objects / tensors that must outlive the lists
        self.tensors = []
        self.graph = None
        self.infer_graph = None
        self.infer_graph_tiles = None     # forward (without the input transpose) + decode, for predict_tiles
        self.zs_ws = None
        self._build()

    # -- allocation helpers -------------------------------------------------
    def _new(self, n, h, w, c, ld=None, zero=False, dtype=torch.float32):
        ld = c if ld is None else ld
        numel = n * h * w * ld
        buf = (torch.zeros if zero else torch.empty)(numel, dtype=dtype, device=self.model.device)
        t = _T(buf, n, h, w, c, ld)
        self.tensors.append(t)      # the launch lists hold raw pointers only: keep every buffer alive with the plan
        return t

    def _emit(self, lst, fn, *args):
        self.keep.append(args)
        lst.append((fn, args))

    def _conv_call(self, lst, m, sp, fn, *args, need=None):
        """Emit a conv launch that may use the shared split-K workspace (allocated after the walk)."""
        if need is None:
            need = int(lib.y3_conv2d_fwd_workspace(m, sp.cin_pad, sp.k, sp.cout))
        self.conv_ws_bytes = max(self.conv_ws_bytes, need)
        entry = [fn, args]
        self._conv_ws_users.append(entry)
        lst.append(entry)

    def _bind_conv_workspace(self):
        # zeroed once: the head of a conv workspace holds per-tile tickets that every launch leaves at zero (yolo3hip.h)
        self.conv_ws = torch.zeros(max(self.conv_ws_bytes // 4, 4), dtype=torch.float32, device=self.model.device)
        for lst in (self.fwd, self.bwd):
            for i, e in enumerate(lst):
                if isinstance(e, list):
                    fn, args = e
                    lst[i] = (fn, tuple(args) + (self.conv_ws.data_ptr(), self.conv_ws_bytes))
        self.keep.append(self._conv_ws_users)

    # -- network walk (model.py:356-421) ---------------------------------------
    def _build(self):
        mdl = self.model
        dev = mdl.device
        N = self.n
        H, W, C = mdl.img_size
        A, K = mdl.number_anchors, mdl.number_classes
        D = A * (5 + K)
        Dld = _round_up(D, 4)
        specs = mdl.specs
        P = mdl.params
        tr = self.training
        bf = self.bf16
        act = torch.bfloat16 if bf else torch.float32      # activation storage type after the first layer
        self.in_nchw = torch.zeros(N, C, H, W, dtype=torch.float32, device=dev)
        self.ops = []      # high-level records for the backward emission
        self.x3_fwd, self.x3_dgrad, self.x3_wgrad = [], [], []     # layers whose forward / data gradient / kernel gradient runs the x3 kernels (bench.py: work per arithmetic)
        self.layer_out = []  # output activation of every conv_layer, creation order (debug / tests)
        li = [0]

        x0 = self._new(N, H, W, specs[0].cin_pad, zero=True)
        self.x0 = x0                      # the network's NHWC input (channels padded to 4): predict_tiles writes it directly
        self._emit(self.fwd, lib.y3_nchw_to_nhwc, self.in_nchw.data_ptr(), N, C, H, W, x0.v)

        # shared workspaces.  The largest M*Cout of the net is conv1's (full resolution, FILTER_COUNT/32
        # channels); BN partial statistics need <= 2*Cout floats per row tile of >= 64 rows.
        max_mc = N * H * W * max(YoloV3.FILTER_COUNT // 32, Dld)
        self.stats_ws = torch.empty(max_mc // 16 + 8192, dtype=torch.float32, device=dev)
        # split-K slabs of the small-spatial layers (y3_conv2d_fwd_workspace / _dgrad_workspace); 64 MiB covers every layer
        # of the 416 / 608 configurations, the exact need is checked per layer below
        self.conv_ws_bytes = 0
        self._conv_ws_users = []
        if not tr:
            # inference-mode BatchNorm (training=False, model.py:38): moving statistics folded into scale / shift, all layers at once
            self._emit(self.fwd, lib.y3_bn_fold_inference_batched, P.data_ptr(), mdl.moving.data_ptr(), mdl.chan.data_ptr(),
                       mdl.fold_table().data_ptr(), mdl.fold_layers, BN_EPS)
        if tr:
            self.dz = torch.empty(max_mc, dtype=torch.float32, device=dev)
            # y3_bn_bwd_workspace(): 1 KiB of tickets (zero before the first launch) + <= 512 x 6 x 64 fp64 partials
            self.bnb_ws = torch.zeros(1024 + 512 * 6 * 64 * 8, dtype=torch.uint8, device=dev)
            self.wg_ws_bytes = 0

        def ptr(off):
            return P.data_ptr() + 4 * off

        def wbf(off):
            return mdl.params_t_bf16.data_ptr() + 2 * off

        def conv_layer(src, out=None, resid=None):
            """model.py:29-39 (+ the tf.add of model.py:47 when resid is given)."""
            i = li[0]
            li[0] += 1
            sp = specs[i]
            oh, ow = -(-src.h // sp.s), -(-src.w // sp.s)
            y = out if out is not None else self._new(N, oh, ow, sp.cout, dtype=act)
            ch = mdl.chan.data_ptr() + 4 * sp.ch_off       # per-layer [scale|shift|mean|rstd|coef(3)] block
            cs = sp.cout * 4
            scale, shift, smean, srstd, coef = ch, ch + cs, ch + 2 * cs, ch + 3 * cs, ch + 4 * cs
            mmean = mdl.moving.data_ptr() + 4 * sp.mv_off
            mvar = mdl.moving.data_ptr() + 4 * (mdl.moving_stride + sp.mv_off)
            # fp32 arithmetic as three bf16 pieces per operand (conv_x3.hip) for the layers the model's policy names: the kernel then
            # wants the copy of the weights with K contiguous per output column, i.e. the TRANSPOSED arena in the forward pass
            x3 = CONV_X3 if (not bf and mdl.x3_forward(sp, N * oh * ow)) else 0
            wfwd = (mdl.planes_t.data_ptr() + 2 * 3 * sp.w_off) if x3 else ptr(sp.w_off)      # x3: the three bf16 planes of the transposed kernel
            fneed = int(lib.y3_conv2d_fwd_workspace_x(N * oh * ow, sp.cin_pad, sp.k, sp.cout, x3))
            if x3:
                self.x3_fwd.append(i)
            if tr:
                a = self._new(N, oh, ow, sp.cout)
                tiles = lib.y3_conv2d_stats_tiles_x(a.m, sp.cin_pad, sp.k, sp.cout, x3)
                assert tiles * 2 * sp.cout <= self.stats_ws.numel()
                self._conv_call(self.fwd, a.m, sp, lib.y3_conv2d_fwd, src.v, wfwd, ptr(sp.b_off), sp.k, sp.s, a.v, EPI_LRELU | x3,
                                LRELU_ALPHA, None, None, None, self.stats_ws.data_ptr(), need=fneed)
                self._emit(self.fwd, lib.y3_bn_stats_finalize, self.stats_ws.data_ptr(), tiles, sp.cout, a.m, ptr(sp.g_off), ptr(sp.be_off),
                           BN_EPS, BN_MOMENTUM, mmean, mvar, smean, srstd, scale, shift)
                self._emit(self.fwd, lib.y3_bn_apply, a.v, scale, shift, resid.v if resid is not None else None, y.v)
                self.ops.append(('conv_layer', i, src, a, y, resid, (smean, srstd, coef)))
            elif bf and i > 0:
                # (the shared conv workspace: small-M layers are split along K, y3_conv2d_fwd_bf16_workspace)
                # the patch kernels of the early 3x3 layers only where the layer streams from HBM (yolo3hip.h, Y3_BF16_NO_PATCH;
                # same box, graph replay: 25 x 608^2 6.54 -> 6.25 ms with them, 8 x 608^2 2.56 -> 2.61, 8 x 416^2 1.74 -> 1.78)
                moved = 2 * y.m * (sp.s * sp.s * sp.cin_pad + sp.cout * (2 if resid is not None else 1))
                self._conv_call(self.fwd, y.m, sp, lib.y3_conv2d_fwd_bf16_ws, src.v, wbf(sp.w_off), ptr(sp.b_off), sp.k, sp.s, y.v, 0,
                                EPI_LRELU | (0 if moved >= BF16_PATCH_MIN_BYTES else BF16_NO_PATCH),
                                LRELU_ALPHA, scale, shift, resid.v if resid is not None else None,
                                need=int(lib.y3_conv2d_fwd_bf16_workspace(y.m, sp.cin_pad, sp.k, sp.cout)))
            elif bf and sp.cin_pad == 4 and sp.cout == 32 and sp.k == 3 and sp.s == 1:
                # the RGB layer: fp32 direct convolution, one rounding on the bf16 store
                self._emit(self.fwd, lib.y3_conv2d_first_bf16, src.v, ptr(sp.w_off), ptr(sp.b_off), y.v, EPI_LRELU, LRELU_ALPHA, scale, shift)
            elif bf:
                # (other first-layer shapes) fp32 MFMA kernel, output rounded to bf16 once
                y32 = self._new(N, oh, ow, sp.cout)
                self._conv_call(self.fwd, y.m, sp, lib.y3_conv2d_fwd, src.v, ptr(sp.w_off), ptr(sp.b_off), sp.k, sp.s, y32.v, EPI_LRELU,
                                LRELU_ALPHA, scale, shift, None, None)
                self._emit(self.fwd, lib.y3_f32_to_bf16, y32.buf.data_ptr(), y.buf.data_ptr(), y.buf.numel())
            else:
                self._conv_call(self.fwd, y.m, sp, lib.y3_conv2d_fwd, src.v, wfwd, ptr(sp.b_off), sp.k, sp.s, y.v, EPI_LRELU | x3,
                                LRELU_ALPHA, scale, shift, resid.v if resid is not None else None, None, need=fneed)
            self.layer_out.append(y)
            return y

        def feature_block(inp, reps, out_last=None):
            """model.py:42-48: every repetition adds the BLOCK input (Q2)."""
            layer = inp
            for r in range(reps):
                layer = conv_layer(layer)
                layer = conv_layer(layer, out=out_last if r == reps - 1 else None, resid=inp)
            return layer

        def yolo_block(inp):
            for _ in range(5):
                inp = conv_layer(inp)
            return inp, conv_layer(inp)

        def head(src, g):
            """detection_layer (model.py:108-120): linear 1x1 conv, bias."""
            i = li[0]
            li[0] += 1
            sp = specs[i]
            fm = self._new(N, src.h, src.w, D, Dld, zero=True)
            if bf:      # bf16 operands, fp32 feature map: decode / loss / NMS stay fp32
                self._conv_call(self.fwd, fm.m, sp, lib.y3_conv2d_fwd_bf16_ws, src.v, wbf(sp.w_off), ptr(sp.b_off), 1, 1, fm.v, 1, 0, 0.0, None, None, None,
                                need=int(lib.y3_conv2d_fwd_bf16_workspace(fm.m, sp.cin_pad, 1, sp.cout)))
                return fm
            self._conv_call(self.fwd, fm.m, sp, lib.y3_conv2d_fwd, src.v, ptr(sp.w_off), ptr(sp.b_off), 1, 1, fm.v, 0, 0.0, None, None, None, None)
            self.ops.append(('head', i, src, fm))
            return fm

        def upsample_into(src, dst):
            self._emit(self.fwd, lib.y3_upsample_sum2x_fwd_bf16 if bf else lib.y3_upsample_sum2x_fwd, src.v, dst.v)
            self.ops.append(('upsample', src, dst))

        FC = YoloV3.FILTER_COUNT
        g1h, g1w = H // 32, W // 32
        cat2 = self._new(N, g1h * 2, g1w * 2, FC, dtype=act)         # tf.concat([up(512), route2(512)])  model.py:368
        cat3 = self._new(N, g1h * 4, g1w * 4, FC // 2, dtype=act)    # tf.concat([up(256), route1(256)])  model.py:375
        cat2_up, cat2_rt = cat2.slice(0, FC // 2), cat2.slice(FC // 2, FC // 2)
        cat3_up, cat3_rt = cat3.slice(0, FC // 4), cat3.slice(FC // 4, FC // 4)

        x = conv_layer(x0)
        x = conv_layer(x)
        x = feature_block(x, 1)
        x = conv_layer(x)
        x = feature_block(x, 2)
        x = conv_layer(x)
        route1 = x = feature_block(x, YoloV3.BLOCK_COUNT, out_last=cat3_rt)
        x = conv_layer(x)
        route2 = x = feature_block(x, YoloV3.BLOCK_COUNT, out_last=cat2_rt)
        x = conv_layer(x)
        route3 = feature_block(x, YoloV3.BLOCK_COUNT // 2)

        route, x = yolo_block(route3)
        fm1 = head(x, 0)
        x = conv_layer(route)
        upsample_into(x, cat2_up)
        route, x = yolo_block(cat2)
        fm2 = head(x, 1)
        x = conv_layer(route)
        upsample_into(x, cat3_up)
        route, x = yolo_block(cat3)
        fm3 = head(x, 2)
        assert li[0] == len(specs)
        self.fms = [fm1, fm2, fm3]
        self.x0 = x0

        # decode (model.py:169-212)
        self.nb = sum(f.h * f.w * A for f in self.fms)
        self.boxes = torch.empty(N, self.nb, 5 + K, dtype=torch.float32, device=dev)
        self.fm_arr = (_hip.Tensor * 3)(*[f.v for f in self.fms])
        self.decode_call = (lib.y3_decode_fwd, (self.fm_arr, 3, mdl.anchors_c, A, K, H, W, self.boxes.data_ptr()))

        # loss (model.py:214-354): always available (test_step needs it in inference mode too)
        self.gt = [torch.zeros(N, f.h, f.w, A, 5 + K, dtype=torch.float32, device=dev) for f in self.fms]
        self.loss4 = torch.zeros(4, dtype=torch.float32, device=dev)
        # one workspace PER SCALE (anchor-present flags + block partials): the three calls of a step then share no word that one
        # clears while another has set or reads it (round 3: a shared flag array cleared by a memset node was mis-ordered in a replayed graph)
        ws_floats = (int(lib.y3_loss_workspace_bytes()) // 4 + 4 + 63) // 64 * 64
        self.loss_ws = torch.zeros(3 * ws_floats, dtype=torch.float32, device=dev)
        self.loss_calls = []
        for si, (f, g) in enumerate(zip(self.fms, self.gt)):
            f.grad = self._new(N, f.h, f.w, D, Dld, zero=True)
            self.loss_calls.append((lib.y3_loss_fwd_bwd, (f.v, g.data_ptr(), mdl.anchors_c, A, K, H, W, float(mdl.global_batch_size),
                                                          self.loss4.data_ptr(), f.grad.v, self.loss_ws.data_ptr() + 4 * si * ws_floats)))
            f.gw = True
        if tr:
            self._build_backward()
        self._bind_conv_workspace()

    # -- backward emission -------------------------------------------------------
    def _grad_of(self, t):
        """Gradient twin of activation t (allocated on first use; slices of a
        concat buffer get slices of the concat's gradient)."""
        if t.grad is None:
            if t.parent is not None:
                self._grad_of(t.parent)
                return t.grad
            t.grad = self._new(t.n, t.h, t.w, t.c, t.ld)
            for ch in t.children:
                ch.grad = _T(t.grad.buf, ch.n, ch.h, ch.w, ch.c, ch.ld, ch.off)
        return t.grad

    def _build_backward(self):
        mdl = self.model
        specs = mdl.specs
        G = mdl.grads
        Wt = mdl.params_t
        N = self.n

        def gptr(off):
            return G.data_ptr() + 4 * off

        # concat parents own the gradient storage of their slices
        wg_need = 0
        for op in self.ops:
            if op[0] in ('conv_layer', 'head'):
                sp = specs[op[1]]
                src = op[2]
                dd = op[3]
                need = lib.y3_conv2d_wgrad_workspace_x(src.v, view(self.dz, dd.n, dd.h, dd.w, sp.cout), sp.k, sp.s,
                                                       CONV_X3 if (op[0] == 'conv_layer' and mdl.x3_wgrad(sp, dd.m)) else 0)
                wg_need = max(wg_need, int(need))
        self.wg_ws = torch.zeros(max(wg_need // 4, 4), dtype=torch.float32, device=mdl.device)      # tickets + slabs, zeroed once
        self.wg_ws_bytes = wg_need

        # BatchNorm-backward statistics without a pass of their own: the gradient dy of a layer's output is complete when the
        # data gradient of its FIRST consumer in forward order has run (later consumers -- residual adds, routes -- are
        # visited earlier by the reversed walk).  If that consumer is a convolution reading exactly this tensor and the
        # shape qualifies (y3_conv2d_dgrad_bn_tiles; stride 2: the merged launch), its data gradient sums the raw moments in its epilogue and
        # the producer only needs y3_bn_bwd_finalize_tiles; every other layer keeps y3_bn_bwd_stats.
        epi_of = {}            # id(consumer op) -> (producer activation a, partial buffer, tiles)
        epi_for = {}           # id(producer op) -> (partial buffer, tiles)
        if BN_EPILOGUE_STATS:
            first_consumer = {}
            for op in self.ops:
                reads = []
                if op[0] == 'conv_layer':
                    reads = [op[2]] + ([op[5]] if op[5] is not None else [])
                elif op[0] == 'head':
                    reads = [op[2]]
                elif op[0] == 'upsample':
                    reads = [op[1]]
                for t in reads:
                    first_consumer.setdefault(id(t), op)
                    if t.parent is not None:
                        first_consumer.setdefault(id(t.parent), op)
                    for ch in t.children:
                        first_consumer.setdefault(id(ch), op)
            for op in self.ops:
                if op[0] != 'conv_layer':
                    continue
                _, i, src, a, y, resid, _ = op
                cons = first_consumer.get(id(y))
                if cons is None or cons[0] != 'conv_layer' or cons[2] is not y or y.children or y is self.x0:
                    continue      # (a concat SLICE qualifies: later readers of the whole concat are visited earlier by the reversed walk)
                if not BN_EPILOGUE_WIDE and (y.parent is not None or specs[cons[1]].s != 1):
                    continue
                csp = specs[cons[1]]
                ca = cons[3]
                dd = view(self.dz, ca.n, ca.h, ca.w, csp.cout)
                tiles = int(lib.y3_conv2d_dgrad_bn_tiles_x(dd, csp.k, csp.s, y.v, CONV_X3 if mdl.x3_dgrad(csp, y.m) else 0))
                if tiles <= 0:
                    continue
                part = torch.empty(tiles * 6 * y.c, dtype=torch.float32, device=mdl.device)
                epi_of[id(cons)] = (a, part, tiles)
                epi_for[id(op)] = (part, tiles)
        self.epilogue_stats_layers = len(epi_for)
        first_src = self.x0
        # The kernel gradient of a layer and its data gradient both start from dz and are independent.  Each is a single
        # round of workgroups with ~8 us of prologue + epilogue in which the matrix pipe idles, so the kernel gradients go to
        # a second stream and fill those bubbles: 23.3 -> 21.4 ms per step with host launches.  (Replayed as a HIP graph
        # the two branches gained nothing -- 23.8 ms -- so a graph-captured step keeps one stream.)  dz is double-buffered:
        # bn_bwd_apply of layer i-2 waits for the kernel gradient of layer i that still reads the buffer.
        two = (not mdl.use_graph) and os.environ.get('Y3_WGRAD_STREAM', '1') != '0'
        self.side = streams.reserve(mdl.device)[0] if two else None      # one per device, bound to its hardware queue early (streams.py)
        self.events = []
        dz_bufs = [self.dz, torch.empty_like(self.dz)] if two else [self.dz]
        dz_busy = [None, None]          # event index of the wgrad still reading each dz buffer
        head_wg = []
        nconv = 0
        for op in reversed(self.ops):
            kind = op[0]
            if kind == 'head':
                _, i, src, fm = op
                sp = specs[i]
                dfm = fm.grad
                self._emit(self.bwd, lib.y3_colsum, dfm.v, gptr(sp.b_off))
                if two:     # every kernel gradient goes through the side stream: they share one slab workspace, in stream order
                    self.events += [torch.cuda.Event(), torch.cuda.Event()]
                    e_go, e_wg = len(self.events) - 2, len(self.events) - 1
                    self.bwd.append(('record', e_go))
                    wargs = (src.v, dfm.v, 1, 1, gptr(sp.w_off), self.wg_ws.data_ptr(), self.wg_ws_bytes)
                    self.keep.append(wargs)
                    self.bwd.append(('side_call', (lib.y3_conv2d_wgrad, wargs, e_go, e_wg)))
                    head_wg.append(e_wg)
                else:
                    self._emit(self.bwd, lib.y3_conv2d_wgrad, src.v, dfm.v, 1, 1, gptr(sp.w_off), self.wg_ws.data_ptr(), self.wg_ws_bytes)
                ds = self._grad_of(src)
                self._conv_call(self.bwd, ds.m, sp, lib.y3_conv2d_dgrad, dfm.v, Wt.data_ptr() + 4 * sp.w_off, 1, 1, ds.v, EPI_ACCUM if src.gw else 0,
                                need=int(lib.y3_conv2d_dgrad_workspace(dfm.v, 1, 1, ds.v)))
                src.mark_written()
                self.bwd.append(('layer_done', i))
            elif kind == 'upsample':
                _, src, dst = op
                assert dst.gw and not src.gw
                ds = self._grad_of(src)
                self._emit(self.bwd, lib.y3_upsample_sum2x_bwd, dst.grad.v, ds.v)
                src.mark_written()
            else:
                _, i, src, a, y, resid, (smean, srstd, coef) = op
                sp = specs[i]
                assert y.gw, 'gradient of layer %d output never produced' % i
                dy = y.grad
                dr = None
                if resid is not None:                       # out = resid + y  ->  d resid += d out, fused into the pass that reads dy anyway
                    dr = self._grad_of(resid)
                    dr_acc = 1 if resid.gw else 0
                    resid.mark_written()
                slot = nconv % len(dz_bufs)
                nconv += 1
                dz = _T(dz_bufs[slot], a.n, a.h, a.w, sp.cout)
                self.keep.append(dz)
                if id(op) in epi_for:
                    # the statistics were summed by the data gradient that completed dy: finalize, then apply (+ residual fan-in)
                    part, tiles = epi_for[id(op)]
                    self._emit(self.bwd, lib.y3_bn_bwd_finalize_tiles, part.data_ptr(), tiles, sp.cout, a.m, mdl.params.data_ptr() + 4 * sp.g_off,
                               smean, srstd, LRELU_ALPHA, gptr(sp.g_off), gptr(sp.be_off), gptr(sp.b_off), coef)
                    if two and dz_busy[slot] is not None:
                        self.bwd.append(('main_wait', dz_busy[slot]))
                    if dr is not None:
                        self._emit(self.bwd, lib.y3_bn_bwd_apply_fanin, dy.v, a.v, coef, LRELU_ALPHA, dz.v, dr.v, dr_acc)
                    else:
                        self._emit(self.bwd, lib.y3_bn_bwd_apply, dy.v, a.v, coef, LRELU_ALPHA, dz.v)
                else:
                    assert int(lib.y3_bn_bwd_workspace(a.m, sp.cout)) <= self.bnb_ws.numel()
                    self._emit(self.bwd, lib.y3_bn_bwd_stats, dy.v, a.v, dr.v if dr is not None else None, dr_acc if dr is not None else 0,
                               mdl.params.data_ptr() + 4 * sp.g_off, smean, srstd, LRELU_ALPHA, gptr(sp.g_off), gptr(sp.be_off), gptr(sp.b_off), coef,
                               self.bnb_ws.data_ptr(), self.bnb_ws.numel())
                    if two and dz_busy[slot] is not None:
                        self.bwd.append(('main_wait', dz_busy[slot]))
                    self._emit(self.bwd, lib.y3_bn_bwd_apply, dy.v, a.v, coef, LRELU_ALPHA, dz.v)
                wx3 = CONV_X3 if mdl.x3_wgrad(sp, a.m) else 0      # kernel gradient on the x3 arithmetic (both operands are activations: no weight copy involved)
                if wx3:
                    self.x3_wgrad.append(i)
                if two:
                    self.events += [torch.cuda.Event(), torch.cuda.Event()]
                    e_dz, e_wg = len(self.events) - 2, len(self.events) - 1
                    self.bwd.append(('record', e_dz))
                    wargs = (src.v, dz.v, sp.k, sp.s, gptr(sp.w_off), wx3, self.wg_ws.data_ptr(), self.wg_ws_bytes)
                    self.keep.append(wargs)
                    self.bwd.append(('side_call', (lib.y3_conv2d_wgrad_x, wargs, e_dz, e_wg)))
                    dz_busy[slot] = e_wg
                else:
                    self._emit(self.bwd, lib.y3_conv2d_wgrad_x, src.v, dz.v, sp.k, sp.s, gptr(sp.w_off), wx3, self.wg_ws.data_ptr(), self.wg_ws_bytes)
                if src is not first_src:
                    ds = self._grad_of(src)
                    # x3 data gradient: the kernel wants K (= this layer's output channels) contiguous per column: the Keras arena
                    x3 = CONV_X3 if mdl.x3_dgrad(sp, ds.m) else 0
                    wdg = (mdl.planes.data_ptr() + 2 * 3 * sp.w_off) if x3 else (Wt.data_ptr() + 4 * sp.w_off)      # x3: planes of the Keras kernel
                    dflags = (EPI_ACCUM if src.gw else 0) | x3
                    dneed = int(lib.y3_conv2d_dgrad_workspace_x(dz.v, sp.k, sp.s, ds.v, x3))
                    if x3:
                        self.x3_dgrad.append(i)
                    if id(op) in epi_of:      # this launch completes d(src): it also sums the BatchNorm-backward moments of the producer
                        pa, part, _ = epi_of[id(op)]
                        self._conv_call(self.bwd, ds.m, sp, lib.y3_conv2d_dgrad_bn, dz.v, wdg, sp.k, sp.s, ds.v, dflags, pa.v, part.data_ptr(), need=dneed)
                        self.keep.append(part)
                    else:
                        self._conv_call(self.bwd, ds.m, sp, lib.y3_conv2d_dgrad, dz.v, wdg, sp.k, sp.s, ds.v, dflags, need=dneed)
                    src.mark_written()
                self.bwd.append(('layer_done', i))
        for e in dz_busy + head_wg[-1:]:
            if two and e is not None:
                self.bwd.append(('main_wait', e))

    # -- execution -------------------------------------------------------------------
    def _run(self, lst, stream, hook=None):
        main = None
        pending = []                   # kernel gradients in flight on the side stream (event indices)
        for fn, args in lst:
            if fn == 'layer_done':
                if hook is not None:
                    dist_ = self.model.dist
                    if pending and (dist_ is None or args in getattr(dist_, '_by_layer', {args: 1})):
                        for e in pending:          # a gradient bucket is about to be all-reduced: its kernel gradients must be in
                            main.wait_event(self.events[e])
                        pending = []
                    hook(args)
                continue
            if fn in ('record', 'main_wait', 'side_call'):
                if main is None:
                    main = torch.cuda.current_stream(self.model.device)
                if fn == 'record':
                    self.events[args].record(main)
                elif fn == 'main_wait':
                    main.wait_event(self.events[args])
                else:
                    f2, a2, e_wait, e_done = args
                    self.side.wait_event(self.events[e_wait])
                    rc = f2(*a2, self.side.cuda_stream)
                    if rc != 0:
                        check(rc, f2.__name__)
                    self.events[e_done].record(self.side)
                    pending.append(e_done)
                continue
            rc = fn(*args, stream)
            if rc != 0:
                check(rc, fn.__name__)

    def run_forward(self, stream, skip_input=False):
        self._run(self.fwd[1:] if skip_input else self.fwd, stream)      # fwd[0] is the NCHW -> NHWC transpose of the input

    def run_decode(self, stream):
        fn, args = self.decode_call
        check(fn(*args, stream), 'y3_decode_fwd')

    def run_loss(self, stream):
        # (a kernel launch, not tensor.zero_(): inside a captured step nothing may turn into a memset node -- DESIGN 9)
        check(lib.y3_fill(self.loss4.data_ptr(), 4, 0.0, stream), 'y3_fill')
        for fn, args in self.loss_calls:
            check(fn(*args, stream), 'y3_loss_fwd_bwd')

    def run_backward(self, stream, hook=None):
        self._run(self.bwd, stream, hook)


class _CallableModel:
    """What get_keras_model()/get_keras_feature_map_model() hand out: callable
    like a Keras model, ``m(batch, training=False)`` (inference.py:58)."""

    def __init__(self, yolo, feature_maps):
        self._y = yolo
        self._fm = feature_maps

    supports_slots = True      # __call__(..., slot=k): k-th independent set of activation buffers (concurrent streams)

    def __call__(self, batch, training=False, slot=0):
        if self._fm:
            return self._y.feature_maps(batch, training=training)
        return self._y.predict(batch, slot=slot)

    def run_tiles(self, img_dev, dtype_code, img_shape, table_ptr, count, tile_size=None, slot=0):
        """inference_tiled's fast path: gather + z-score + forward + decode of `count` tiles (YoloV3.predict_tiles)."""
        return self._y.predict_tiles(img_dev, dtype_code, img_shape, table_ptr, count, tile_size=tile_size, slot=slot)

    @property
    def trainable_weights(self):
        return self._y.trainable_weights()


class YoloV3:
    # Constants controlling the network (model.py:22-26)
    BLOCK_COUNT = 8
    FILTER_COUNT = 1024
    KERNEL_SIZE = 3
    NETWORK_DOWNSAMPLE_FACTOR = 32
    WEIGHT_DECAY = 5e-4      # declared by the reference but never applied (Q9)

    def __init__(self, global_batch_size, img_size, number_classes, anchors=None, learning_rate=1e-4, device=None, seed=None,
                 use_graph=False, inference_precision='fp32', conv_arithmetic=None):
        if not torch.cuda.is_available():
            raise RuntimeError('yolo3.model.YoloV3 needs an MI355X (HIP) device: there is no CPU path')
        self.device = torch.device(device if device is not None else 'cuda:%d' % torch.cuda.current_device())
        self.number_classes = int(number_classes)
        self.learning_rate = float(learning_rate)
        self.global_batch_size = global_batch_size
        self.img_size = [int(v) for v in img_size]           # [H, W, C]  (model.py:428,440)
        if self.img_size[0] % 32 or self.img_size[1] % 32:
            raise ValueError('image size must be a multiple of %d' % YoloV3.NETWORK_DOWNSAMPLE_FACTOR)
        self.score_threshold = 0.1                           # dead attributes kept (Q10)
        self.iou_threshold = 0.5
        self.anchors = [(32, 32), (128, 128), (256, 256)] if anchors is None else [tuple(a) for a in anchors]
        self.number_anchors = len(self.anchors)
        self.anchors_c = _hip.float_array([v for a in self.anchors for v in a])
        H, W, C = self.img_size
        f = YoloV3.NETWORK_DOWNSAMPLE_FACTOR
        self.box_count_fm1 = (H / f) * (W / f)
        self.box_count_fm2 = (H / (f / 2)) * (W / (f / 2))
        self.box_count_fm3 = (H / (f / 4)) * (W / (f / 4))
        self.number_output_boxes = self.number_anchors * (self.box_count_fm1 + self.box_count_fm2 + self.box_count_fm3)
        self.output_shape = [self.number_output_boxes, 5 + self.number_classes]

        self.specs, self.arena_floats, chan_floats, self.moving_stride = build_layer_specs(C, self.number_anchors, self.number_classes)
        dev = self.device
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
        self.params = z(self.arena_floats)        # trainable arena: [W | b | gamma | beta] per layer
        self.params_t = z(self.arena_floats)      # kernels with channel axes swapped (dgrad operand)
        self.grads = z(self.arena_floats)
        self.adam_m = z(self.arena_floats)
        self.adam_v = z(self.arena_floats)
        self.moving = z(2 * self.moving_stride)   # [moving_mean | moving_var]
        self.chan = z(chan_floats)                # per BN layer: scale | shift | save_mean | save_rstd | k1 | k2 | k3
        self.lr_t_dev = z(1)
        self.beta1, self.beta2, self.adam_eps = 0.9, 0.999, 1e-7   # Keras Adam defaults (App. C5)
        self.iterations = 0
        self.use_graph = bool(use_graph)
        if inference_precision not in ('fp32', 'bf16'):
            raise ValueError("inference_precision must be 'fp32' or 'bf16'")
        self.inference_precision = inference_precision   # predict() default; training is always fp32
        # fp32 convolutions: 'f32' = v_mfma_f32_32x32x2_f32 everywhere; 'x3' = the layers named by x3_forward / x3_dgrad run their
        # fp32 arithmetic as three bf16 pieces per operand on the bf16 matrix pipe (Y3_CONV_X3, conv_x3.hip: fp32-class results,
        # 2.67x fewer matrix-pipe cycles).  Default 'x3'; environment Y3_CONV_X3=0 / 1 / all overrides the default.
        if conv_arithmetic is None:
            conv_arithmetic = {'0': 'f32', '1': 'x3', 'all': 'x3-all'}.get(os.environ.get('Y3_CONV_X3', '1'), 'x3')
        if conv_arithmetic not in ('f32', 'x3', 'x3-all'):
            raise ValueError("conv_arithmetic must be 'f32' or 'x3'")
        self.conv_arithmetic = conv_arithmetic
        # the x3 kernels read their weights as three bf16 piece planes (y3_x3_split_weights) of the copy with K contiguous per output
        # column: of params_t in the forward pass, of params in the data gradient; refreshed with params_t after every optimiser step
        self.planes = self.planes_t = None
        if conv_arithmetic != 'f32':
            self.planes = torch.zeros(3 * self.arena_floats, dtype=torch.bfloat16, device=dev)
            self.planes_t = torch.zeros(3 * self.arena_floats, dtype=torch.bfloat16, device=dev)
        self._x3_table = None
        self.params_t_bf16 = None                 # bf16 copy of params_t, made on first bf16 predict
        self._bf16_stale = True
        self.dist = None                          # set by parallel.DataParallel.attach()
        self._plans = {}
        self._tr_table = None
        self._fold_table = None
        self._init_weights(seed)
        self.model = _CallableModel(self, False)
        self.model_feature_maps = _CallableModel(self, True)
        self.optimizer = self

    # ---- which launches run the x3 kernels ---------------------------------------
    def _x3_policy(self, c, ntaps, nout, m, stride, forward):
        if self.conv_arithmetic == 'f32' or not lib.y3_conv2d_x3_ok(m, c, ntaps, nout) or (stride != 1 and not forward):
            return False
        if self.conv_arithmetic == 'x3-all':
            return True
        # measured (tools/x3_check.py --all, tools/layer_times.py; batch 8 at 416^2, launch by launch and inside the step): the 3x3
        # layers gain -- forward 1.2-1.8x, stride-1 data gradients 1.1-1.7x with >= 128 contracted channels and >= 64 outputs
        # (0.79x for the 64 -> 32 one); the 1x1 layers (4-64 K steps: prologue + epilogue bound) do not
        if ntaps != 9:
            return False
        if forward:
            return True          # every 3x3 layer the kernels take (>= 32 input channels): 208^2 32->64 140 -> 116 us, the others 1.5-1.8x
        return c >= 128 and nout >= 64

    def x3_forward(self, sp, m_out):
        return self._x3_policy(sp.cin_pad, sp.k * sp.k, sp.cout, m_out, sp.s, True)

    def x3_dgrad(self, sp, m_in):
        if sp.s == 2:
            # the merged launch of the four parity classes (y3_conv2d_dgrad_x3_ok): >= 64 input channels, output channels a power of two
            if self.conv_arithmetic == 'f32' or sp.k != 3 or sp.cin_pad < 64 or sp.cout % 16 or (sp.cout & (sp.cout - 1)):
                return False
            # measured (tools/layer_times.py): 163 -> 125, 151 -> 124, 142 -> 103 us at 13^2 / 26^2 / 52^2 output; 152 -> 151 for the 64-channel layer
            return self.conv_arithmetic == 'x3-all' or sp.cin_pad >= 128
        return self._x3_policy(sp.cout, sp.k * sp.k, sp.cin_pad, m_in, sp.s, False)

    def x3_wgrad(self, sp, m_out):
        if self.conv_arithmetic == 'f32' or not lib.y3_conv2d_wgrad_x3_ok(m_out, sp.cin_pad, sp.k, sp.cout):
            return False
        # measured (tools/x3_check.py --wgrad): every 3x3 layer with >= 128 output channels 1.3-1.7x, stride 2 included; the 1x1
        # layers 1.1-1.2x at 52^2 / 26^2 and slower at 13^2
        return self.conv_arithmetic == 'x3-all' or sp.k == 3 or m_out >= 5000

    # ---- construction helpers --------------------------------------------------
    def _init_weights(self, seed):
        """Keras defaults: Glorot-uniform kernels, zero bias, gamma 1, beta 0,
        moving mean 0 / variance 1 (App. C2, C4)."""
        rng = np.random.default_rng(seed)
        layers = []
        for sp in self.specs:
            limit = math.sqrt(6.0 / (sp.k * sp.k * sp.cin + sp.k * sp.k * sp.cout))
            d = dict(W=rng.uniform(-limit, limit, (sp.k, sp.k, sp.cin, sp.cout)).astype(np.float32), b=np.zeros(sp.cout, np.float32))
            if sp.bn:
                d.update(gamma=np.ones(sp.cout, np.float32), beta=np.zeros(sp.cout, np.float32), mean=np.zeros(sp.cout, np.float32),
                         var=np.ones(sp.cout, np.float32))
            layers.append(d)
        self.set_weights(layers)

    # ---- weights in / out (Keras shapes, true Cin) --------------------------------------
    def set_weights(self, layers):
        """layers: list (creation order) of dicts W[kh,kw,Cin,Cout], b, and for
        conv_layers gamma, beta, mean, var."""
        assert len(layers) == len(self.specs)
        host = np.zeros(self.arena_floats, np.float32)
        mov = np.zeros(2 * self.moving_stride, np.float32)
        mov[self.moving_stride:] = 1.0
        for sp, d in zip(self.specs, layers):
            Wk = np.asarray(d['W'], np.float32)
            assert Wk.shape == (sp.k, sp.k, sp.cin, sp.cout), (Wk.shape, (sp.k, sp.k, sp.cin, sp.cout))
            Wp = np.zeros((sp.k, sp.k, sp.cin_pad, sp.cout), np.float32)
            Wp[:, :, :sp.cin, :] = Wk
            host[sp.w_off:sp.w_off + Wp.size] = Wp.ravel()
            host[sp.b_off:sp.b_off + sp.cout] = np.asarray(d['b'], np.float32)
            if sp.bn:
                host[sp.g_off:sp.g_off + sp.cout] = np.asarray(d['gamma'], np.float32)
                host[sp.be_off:sp.be_off + sp.cout] = np.asarray(d['beta'], np.float32)
                mov[sp.mv_off:sp.mv_off + sp.cout] = np.asarray(d['mean'], np.float32)
                mov[self.moving_stride + sp.mv_off:self.moving_stride + sp.mv_off + sp.cout] = np.asarray(d['var'], np.float32)
        self.params.copy_(torch.from_numpy(host))
        self.moving.copy_(torch.from_numpy(mov))
        self._refresh_transposed()

    def _unpack(self, arena):
        host = arena.detach().cpu().numpy()
        out = []
        for sp in self.specs:
            Wp = host[sp.w_off:sp.w_off + sp.k * sp.k * sp.cin_pad * sp.cout].reshape(sp.k, sp.k, sp.cin_pad, sp.cout)
            d = dict(W=Wp[:, :, :sp.cin, :].copy(), b=host[sp.b_off:sp.b_off + sp.cout].copy())
            if sp.bn:
                d['gamma'] = host[sp.g_off:sp.g_off + sp.cout].copy()
                d['beta'] = host[sp.be_off:sp.be_off + sp.cout].copy()
            out.append(d)
        return out

    def get_weights(self, moving=None):
        """moving: optional replacement for self.moving (the cross-replica MEAN a checkpoint stores, parallel.mean_moving_stats)."""
        out = self._unpack(self.params)
        mov = (self.moving if moving is None else moving).detach().cpu().numpy()
        for sp, d in zip(self.specs, out):
            if sp.bn:
                d['mean'] = mov[sp.mv_off:sp.mv_off + sp.cout].copy()
                d['var'] = mov[self.moving_stride + sp.mv_off:self.moving_stride + sp.mv_off + sp.cout].copy()
        return out

    def get_gradients(self):
        """Last step's gradients, same structure as get_weights() (W, b, gamma, beta)."""
        return self._unpack(self.grads)

    def trainable_weights(self):
        """Flat list of arrays in Keras trainable_weights order (model.py:496)."""
        out = []
        for sp, d in zip(self.specs, self._unpack(self.params)):
            out += [d['W'], d['b']] + ([d['gamma'], d['beta']] if sp.bn else [])
        return out

    def save_weights(self, path, moving=None):
        """Own weight file (the reference's TF checkpoint / SavedModel formats need TF)."""
        flat = {}
        for i, d in enumerate(self.get_weights(moving)):
            for k, v in d.items():
                flat['l%03d_%s' % (i, k)] = v
        flat['meta_img_size'] = np.asarray(self.img_size, np.int64)
        flat['meta_number_classes'] = np.asarray(self.number_classes, np.int64)
        flat['meta_anchors'] = np.asarray(self.anchors, np.float32)
        flat['meta_global_batch_size'] = np.asarray(self.global_batch_size, np.int64)
        flat['meta_learning_rate'] = np.asarray(self.learning_rate, np.float64)
        flat['opt_iterations'] = np.asarray(self.iterations, np.int64)
        flat['opt_m'] = self.adam_m.detach().cpu().numpy()
        flat['opt_v'] = self.adam_v.detach().cpu().numpy()
        np.savez(path, **flat)

    def load_weights(self, path, load_optimizer=False):
        z = np.load(path, allow_pickle=False)
        layers = []
        for i, sp in enumerate(self.specs):
            d = {k: z['l%03d_%s' % (i, k)] for k in (['W', 'b', 'gamma', 'beta', 'mean', 'var'] if sp.bn else ['W', 'b'])}
            layers.append(d)
        self.set_weights(layers)
        if load_optimizer and 'opt_m' in z:
            self.adam_m.copy_(torch.from_numpy(z['opt_m']))
            self.adam_v.copy_(torch.from_numpy(z['opt_v']))
            self.iterations = int(z['opt_iterations'])

    @staticmethod
    def from_file(path, device=None):
        z = np.load(path, allow_pickle=False)
        y = YoloV3(int(z['meta_global_batch_size']), [int(v) for v in z['meta_img_size']], int(z['meta_number_classes']),
                   [tuple(float(v) for v in a) for a in z['meta_anchors']], float(z['meta_learning_rate']), device=device)
        y.load_weights(path)
        return y

    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _refresh_transposed(self):
        """params_t <- kernels with the channel axes swapped (operand of the data gradient), all layers in one launch."""
        if self._tr_table is None:
            rows, start = [], 0
            for sp in self.specs[1:]:          # the first layer has no data gradient
                rows.append([sp.w_off, sp.k * sp.k, sp.cin_pad, sp.cout, start])
                start += sp.k * sp.k * (-(-sp.cin_pad // 32)) * (-(-sp.cout // 32))
            self._tr_table = torch.tensor(rows, dtype=torch.int32, device=self.device)
            self._tr_tiles = start
        if self.planes is not None and os.environ.get('Y3_PREP_FUSED', '1') != '0':
            # one pass over the arena: transposed copy + the piece planes of both copies (bit-identical to the three launches below)
            check(lib.y3_x3_prepare_weights_batched(self.params.data_ptr(), self.params_t.data_ptr(), self.planes.data_ptr(), self.planes_t.data_ptr(),
                                                    self._tr_table.data_ptr(), len(self.specs) - 1, self._tr_tiles, self._stream()), 'y3_x3_prepare_weights_batched')
            self._bf16_stale = True
            return
        check(lib.y3_transpose_weights_batched(self.params.data_ptr(), self.params_t.data_ptr(), self._tr_table.data_ptr(), len(self.specs) - 1,
                                               self._tr_tiles, self._stream()), 'y3_transpose_weights_batched')
        if self.planes is not None:
            if self._x3_table is None:
                # the layers whose channel counts the x3 kernels take (K per row a multiple of 16), per copy: {offset, taps, rows, K per row, first block}
                tabs = []
                for fwd_copy in (False, True):
                    rows, start = [], 0
                    for sp in self.specs[1:]:
                        nrows, kpr = (sp.cout, sp.cin_pad) if fwd_copy else (sp.cin_pad, sp.cout)      # params_t: [tap][Cout][Cin]; params: [tap][Cin][Cout]
                        if kpr % 16:
                            continue
                        rows.append([sp.w_off, sp.k * sp.k, nrows, kpr, start])
                        start += -(-(sp.k * sp.k * nrows * kpr) // 1024)
                    tabs.append((torch.tensor(rows, dtype=torch.int32, device=self.device), len(rows), start))
                self._x3_table = tabs
            for (arena, planes), (tab, nl, blocks) in zip(((self.params, self.planes), (self.params_t, self.planes_t)), self._x3_table):
                check(lib.y3_x3_split_weights_batched(arena.data_ptr(), planes.data_ptr(), tab.data_ptr(), nl, blocks, self._stream()), 'y3_x3_split_weights_batched')
        self._bf16_stale = True

    def fold_table(self):
        """Device table for y3_bn_fold_inference_batched (one row per BatchNorm layer)."""
        if self._fold_table is None:
            rows = [[sp.g_off, sp.be_off, sp.mv_off, self.moving_stride + sp.mv_off, sp.ch_off, sp.ch_off + sp.cout, sp.cout]
                    for sp in self.specs if sp.bn]
            self.fold_layers = len(rows)
            self._fold_table = torch.tensor(rows, dtype=torch.int32, device=self.device)
        return self._fold_table

    def _refresh_bf16(self):
        """params_t_bf16 <- round-to-nearest-even of params_t (the [tap][Cout][Cin] operand of y3_conv2d_fwd_bf16)."""
        if self.params_t_bf16 is None:
            self.params_t_bf16 = torch.zeros(self.arena_floats, dtype=torch.bfloat16, device=self.device)
        if self._bf16_stale:
            check(lib.y3_f32_to_bf16(self.params_t.data_ptr(), self.params_t_bf16.data_ptr(), self.arena_floats, self._stream()), 'y3_f32_to_bf16')
            torch.cuda.current_stream(self.device).synchronize()     # rare (weights changed); other streams may read the copy next
            self._bf16_stale = False

    # ---- reference API (model.py:466-479) ---------------------------------------------
    def get_keras_model(self):
        return self.model

    def get_keras_feature_map_model(self):
        return self.model_feature_maps

    def get_optimizer(self):
        return self.optimizer

    def set_learning_rate(self, learning_rate):
        self.learning_rate = float(learning_rate)

    def get_learning_rate(self):
        return self.learning_rate

    # ---- execution -------------------------------------------------------------------------
    def _plan(self, n, training, bf16=False, slot=0):
        bf16 = bool(bf16) and not training
        key = (int(n), bool(training), bf16, int(slot))
        if bf16:
            self._refresh_bf16()
        if key not in self._plans:
            self._plans[key] = _Plan(self, int(n), bool(training), bf16)
        return self._plans[key]

    def _load_inputs(self, plan, images, gt_data=None):
        images = torch.as_tensor(images)
        if tuple(images.shape[1:]) != (self.img_size[2], self.img_size[0], self.img_size[1]):
            raise ValueError('input shape %s does not match the model input (C,H,W)=%s (Q18: fixed at construction)'
                             % (tuple(images.shape), (self.img_size[2], self.img_size[0], self.img_size[1])))
        plan.in_nchw.copy_(images.to(torch.float32), non_blocking=True)
        if gt_data is not None:
            for dst, src in zip(plan.gt, gt_data):
                dst.copy_(torch.as_tensor(src).to(torch.float32).reshape(dst.shape), non_blocking=True)

    def predict(self, images, precision=None, slot=0):
        """The saved 'yolov3' model (model.py:463): NCHW in -> [N, Nb, 5+K].  precision 'bf16' runs every conv after
        the RGB layer on the bf16 MFMA path (fp32 accumulate, fp32 heads / decode); default self.inference_precision.
        Calls with different ``slot`` use separate activation / output buffers and may run concurrently on different
        streams (inference_tiled does that); calls with the same slot must be stream-ordered."""
        n = int(images.shape[0])
        plan = self._plan(n, False, (precision or self.inference_precision) == 'bf16', slot)
        self._load_inputs(plan, images)
        if self.use_graph:
            if plan.infer_graph is None:
                self._capture_inference(plan)
            plan.infer_graph.replay()
        else:
            st = self._stream()
            plan.run_forward(st)
            plan.run_decode(st)
        return plan.boxes

    def predict_tiles(self, img_dev, dtype_code, img_shape, table_ptr, count, tile_size=None, precision=None, slot=0):
        """One batch of inference_tiled: tiles `table_ptr[0:count]` (device rows of tile_table) of the device-resident HWC image
        -> z-scored NHWC network input in two passes over the image (y3_tile_gather_zscore_nhwc; the same bits as
        tiles_to_device -> zscore_normalize_device -> the input transpose), then forward + decode.  Returns [count, Nb, 5+K]."""
        h, w, c = (int(v) for v in img_shape)
        if c != self.img_size[2] or (tile_size is not None and [int(tile_size[0]), int(tile_size[1])] != self.img_size[:2]):
            raise ValueError('tiles %s x %d channels do not match the model input (H,W,C)=%s (Q18: fixed at construction)' % (tile_size, c, self.img_size))
        count = int(count)
        plan = self._plan(count, False, (precision or self.inference_precision) == 'bf16', slot)
        st = self._stream()
        if plan.zs_ws is None:
            plan.zs_ws = torch.empty(int(lib.y3_zscore_workspace_bytes(count)) // 8 + 1, dtype=torch.float64, device=self.device)
        check(lib.y3_tile_gather_zscore_nhwc(img_dev.data_ptr(), int(dtype_code), h, w, c, table_ptr, count, self.img_size[0], self.img_size[1],
                                             plan.x0.buf.data_ptr(), plan.x0.ld, plan.zs_ws.data_ptr(), st), 'y3_tile_gather_zscore_nhwc')
        if self.use_graph:
            if plan.infer_graph_tiles is None:
                plan.run_forward(st, skip_input=True)
                plan.run_decode(st)
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode='thread_local'):
                    st2 = self._stream()
                    plan.run_forward(st2, skip_input=True)
                    plan.run_decode(st2)
                plan.infer_graph_tiles = g
            plan.infer_graph_tiles.replay()
        else:
            plan.run_forward(st, skip_input=True)
            plan.run_decode(st)
        return plan.boxes

    def _capture_inference(self, plan):
        """forward + decode of one inference plan as a HIP graph (the launch lists are static and read no host state)."""
        st = self._stream()
        plan.run_forward(st)
        plan.run_decode(st)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):      # the prefetch thread may allocate pinned memory meanwhile
            st = self._stream()
            plan.run_forward(st)
            plan.run_decode(st)
        plan.infer_graph = g

    def feature_maps(self, images, training=False, precision=None):
        """The 'yolov3_fm' model (model.py:462): three NCHW feature maps (fp32 unless precision='bf16' is asked for)."""
        n = int(images.shape[0])
        plan = self._plan(n, training, precision == 'bf16')
        self._load_inputs(plan, images)
        st = self._stream()
        plan.run_forward(st)
        return self._export_fms(plan, st)

    def _export_fms(self, plan, st):
        out = []
        for f in plan.fms:
            t = torch.empty(f.n, f.c, f.h, f.w, dtype=torch.float32, device=self.device)
            check(lib.y3_nhwc_to_nchw(f.v, t.data_ptr(), st), 'y3_nhwc_to_nchw')
            out.append(t)
        return out

    def _lr_t(self):
        t = self.iterations
        return self.learning_rate * math.sqrt(1.0 - self.beta2 ** t) / (1.0 - self.beta1 ** t)

    def _fwd_bwd(self, plan, st):
        plan.run_forward(st)
        plan.run_loss(st)
        hook = self.dist.on_layer_done if self.dist is not None else None
        if self.dist is not None:
            self.dist.begin_step()
        plan.run_backward(st, hook)

    def _adam(self, st):
        check(lib.y3_adam_step(self.params.data_ptr(), self.grads.data_ptr(), self.adam_m.data_ptr(), self.adam_v.data_ptr(),
                               self.arena_floats, self.lr_t_dev.data_ptr(), self.beta1, self.beta2, self.adam_eps, st), 'y3_adam_step')
        self._refresh_transposed()

    def train_step(self, inputs):
        """model.py:481-508 for this replica.  inputs = (images, (gt1, gt2, gt3),
        loss_metric, loss_xy_metric, loss_wh_metric, loss_obj_metric, loss_class_metric);
        metrics may be None.  Returns the loss value as a 0-d device tensor."""
        images, gt_data = inputs[0], inputs[1]
        metrics = list(inputs[2:]) + [None] * 5
        n = int(images.shape[0])
        plan = self._plan(n, True)
        self._load_inputs(plan, images, gt_data)
        self.iterations += 1
        self._bf16_stale = True
        self.lr_t_dev.fill_(self._lr_t())
        st = self._stream()
        if self.use_graph and self.dist is None:
            if plan.graph is None:
                self._capture(plan)
            plan.graph.replay()
        else:
            self._fwd_bwd(plan, st)
            if self.dist is not None:
                self.dist.finish_step()
            self._adam(st)
        parts = plan.loss4.clone()
        loss_value = parts.sum() / float(self.global_batch_size)
        for mtr, val in zip(metrics[:5], [loss_value, parts[0], parts[1], parts[2], parts[3]]):
            if mtr is not None:
                mtr.update_state(val)
        return loss_value

    def _capture(self, plan):
        """Capture forward + loss + backward + Adam of one step into a HIP graph.
        A warm-up pass (no Adam) runs first so lazy initialisation happens
        outside the capture; it only touches scratch state plus the BN moving
        statistics, which are restored before capturing."""
        moving = self.moving.clone()
        st = self._stream()
        plan.run_forward(st)
        plan.run_loss(st)
        plan.run_backward(st)
        self.moving.copy_(moving)
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode='thread_local'):      # the prefetch thread may allocate pinned memory meanwhile
            st = self._stream()
            plan.run_forward(st)
            plan.run_loss(st)
            plan.run_backward(st)
            self._adam(st)
        plan.graph = g

    def dist_train_step(self, dist_strategy, inputs):
        """model.py:510-515: per-replica step + SUM of the per-replica losses."""
        if dist_strategy is not None and self.dist is None:
            dist_strategy.attach(self)
        loss = self.train_step(inputs)
        return dist_strategy.reduce_sum(loss) if dist_strategy is not None else loss

    def test_step(self, inputs):
        """model.py:517-534: BN in inference mode, loss + metrics, no update."""
        images, gt_data = inputs[0], inputs[1]
        metrics = list(inputs[2:]) + [None] * 5
        n = int(images.shape[0])
        plan = self._plan(n, False)
        self._load_inputs(plan, images, gt_data)
        st = self._stream()
        plan.run_forward(st)
        plan.run_loss(st)
        parts = plan.loss4.clone()
        loss_value = parts.sum() / float(self.global_batch_size)
        for mtr, val in zip(metrics[:5], [loss_value, parts[0], parts[1], parts[2], parts[3]]):
            if mtr is not None:
                mtr.update_state(val)
        return loss_value

    def dist_test_step(self, dist_strategy, inputs):
        """model.py:536-540."""
        loss = self.test_step(inputs)
        return dist_strategy.reduce_sum(loss) if dist_strategy is not None else loss
