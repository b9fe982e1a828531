"""Oracle (test infrastructure): torch-CPU restatement of the reference's
TensorFlow model -- network, anchor decode, loss, train step, Keras Adam.

PARITY UNPINNED: the reference (/root/reference/model.py) runs on TensorFlow
2.0/2.1, which is absent here, and ships no tests/fixtures for this path.
Every function cites the model.py lines it restates; TF op semantics used are
listed in SURVEY.md App. C (SAME padding, leaky_relu alpha 0.2, BN eps 1e-3 /
momentum .99, Keras Adam eps 1e-7, sigmoid-CE formula, clip gradient).

Layout here is the reference's own: NCHW activations, Keras kernels
[kh, kw, Cin, Cout].  dtype is selectable (float32 to mirror the reference,
float64 to bound rounding error in tests).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

NETWORK_DOWNSAMPLE_FACTOR = 32   # model.py:25
BLOCK_COUNT = 8                  # model.py:22
FILTER_COUNT = 1024              # model.py:23
BN_EPS = 1e-3                    # Keras BatchNormalization default (App. C4)
BN_MOMENTUM = 0.99
LRELU_ALPHA = 0.2                # tf.nn.leaky_relu default (App. C3)


# ----------------------------------------------------------------------------
# architecture walk (model.py:356-421) -> ordered layer specs
# ----------------------------------------------------------------------------
def layer_specs(in_channels, num_anchors, num_classes):
    """Ordered conv specs in Keras creation order (= trainable_weights order).
    Each: dict(cin, cout, k, s, bn(bool))."""
    L = []

    def conv(cin, cout, k, s=1, bn=True):
        L.append(dict(cin=cin, cout=cout, k=k, s=s, bn=bn))
        return cout

    def feature_block(c, reps):            # model.py:42-48
        for _ in range(reps):
            conv(c, c // 2, 1)
            conv(c // 2, c, 3)
        return c

    c = conv(in_channels, FILTER_COUNT // 32, 3)        # :385
    c = conv(c, FILTER_COUNT // 16, 3, 2)               # :387
    c = feature_block(c, 1)                             # :390
    c = conv(c, FILTER_COUNT // 8, 3, 2)                # :393
    c = feature_block(c, 2)                             # :396
    c = conv(c, FILTER_COUNT // 4, 3, 2)                # :399
    c = feature_block(c, BLOCK_COUNT)                   # :402  route1 (256)
    c = conv(c, FILTER_COUNT // 2, 3, 2)                # :406
    c = feature_block(c, BLOCK_COUNT)                   # :409  route2 (512)
    c = conv(c, FILTER_COUNT, 3, 2)                     # :413
    c = feature_block(c, BLOCK_COUNT // 2)              # :416  route3 (1024)
    D = num_anchors * (5 + num_classes)

    def yolo_block(cin, fc):                            # model.py:51-59
        conv(cin, fc // 2, 1)
        conv(fc // 2, fc, 3)
        conv(fc, fc // 2, 1)
        conv(fc // 2, fc, 3)
        conv(fc, fc // 2, 1)
        conv(fc // 2, fc, 3)

    yolo_block(1024, 1024)                              # :363
    conv(1024, D, 1, 1, bn=False)                       # :364 feature_map_1
    conv(512, 512, 1)                                   # :366 lateral (Q4)
    yolo_block(1024, 512)                               # :370
    conv(512, D, 1, 1, bn=False)                        # :371
    conv(256, 256, 1)                                   # :373
    yolo_block(512, 256)                                # :377
    conv(256, D, 1, 1, bn=False)                        # :378
    return L


def init_params(in_channels, num_anchors, num_classes, seed=1, randomize_bn=False, dtype=np.float32):
    """Keras defaults (App. C2/C4): Glorot-uniform kernels, zero bias, gamma 1,
    beta 0, moving mean 0 / var 1.  ``randomize_bn`` perturbs BN so inference
    epilogues are exercised."""
    rng = np.random.default_rng(seed)
    P = []
    for sp in layer_specs(in_channels, num_anchors, num_classes):
        k, cin, cout = sp['k'], sp['cin'], sp['cout']
        limit = math.sqrt(6.0 / (k * k * cin + k * k * cout))
        p = dict(W=rng.uniform(-limit, limit, (k, k, cin, cout)).astype(dtype),
                 b=np.zeros(cout, dtype))
        if sp['bn']:
            if randomize_bn:
                p.update(gamma=rng.uniform(0.5, 1.5, cout).astype(dtype),
                         beta=rng.normal(0, 0.1, cout).astype(dtype),
                         mean=rng.normal(0, 0.1, cout).astype(dtype),
                         var=rng.uniform(0.5, 1.5, cout).astype(dtype))
                p['b'] = rng.normal(0, 0.1, cout).astype(dtype)
            else:
                p.update(gamma=np.ones(cout, dtype), beta=np.zeros(cout, dtype),
                         mean=np.zeros(cout, dtype), var=np.ones(cout, dtype))
        P.append(p)
    return P


def same_pad(size, k, s):
    """TF padding='same' (App. C1): (pad_before, pad_after)."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return total // 2, total - total // 2


# ----------------------------------------------------------------------------
# forward
# ----------------------------------------------------------------------------
class Net:
    """Functional network over a list of per-layer tensors."""

    def __init__(self, params, in_channels, num_anchors, num_classes, dtype=torch.float32, requires_grad=False):
        self.specs = layer_specs(in_channels, num_anchors, num_classes)
        self.dtype = dtype
        self.p = []
        for sp, p in zip(self.specs, params):
            q = {}
            for name, v in p.items():
                t = torch.tensor(np.asarray(v), dtype=dtype)
                if requires_grad and name in ('W', 'b', 'gamma', 'beta'):
                    t.requires_grad_(True)
                q[name] = t
            self.p.append(q)
        self.batch_stats = []   # (mean, biased var, count) per BN layer of the last training fwd
        self.trace = None       # optional list of per-layer outputs (NCHW)
        self.force = None       # optional teacher forcing: {'layers': [NCHW per conv_layer], 'up': [NCHW per upsample]} --
                                # every layer is still computed (and traced) from its inputs, but the NEXT layer consumes
                                # the forced tensor, so a per-layer comparison does not compound differences
        self.trace_exact = None  # like trace, but BEFORE the bf16 rounding (identical to trace when bf16 is off)
        self.up_trace = None    # optional list of the two upsample outputs (before rounding)
        self.bf16 = False       # emulate the bf16 inference path: operands of every conv after the first rounded to
                                # bf16 (round-to-nearest-even), fp32 accumulate / epilogue, one rounding per stored activation

    def _r(self, t):
        return t.to(torch.bfloat16).to(t.dtype) if self.bf16 else t

    def trainable(self):
        out = []
        for sp, q in zip(self.specs, self.p):
            out += [q['W'], q['b']]
            if sp['bn']:
                out += [q['gamma'], q['beta']]
        return out

    def _conv_layer(self, x, i, training):
        """model.py:29-39: conv(+bias, SAME) -> leaky_relu(0.2) -> BatchNorm."""
        sp, q = self.specs[i], self.p[i]
        k, s = sp['k'], sp['s']
        ph = same_pad(x.shape[2], k, s)
        pw = same_pad(x.shape[3], k, s)
        x = F.pad(x, (pw[0], pw[1], ph[0], ph[1]))
        w = q['W'].permute(3, 2, 0, 1)            # [kh,kw,Cin,Cout] -> [Cout,Cin,kh,kw]
        if self.bf16 and i > 0:
            w = self._r(w)
        z = F.conv2d(x, w, q['b'], stride=s)
        if not sp['bn']:
            return z                               # detection_layer model.py:108-120 (linear)
        a = F.leaky_relu(z, LRELU_ALPHA)
        if training:
            mean = a.mean(dim=(0, 2, 3))
            var = a.var(dim=(0, 2, 3), unbiased=False)
            self.batch_stats.append((mean.detach(), var.detach(), a.numel() // a.shape[1]))
        else:
            mean, var = q['mean'], q['var']
        y = (a - mean[None, :, None, None]) * torch.rsqrt(var[None, :, None, None] + BN_EPS)
        return y * q['gamma'][None, :, None, None] + q['beta'][None, :, None, None]

    @staticmethod
    def _upsample_2x(x):
        """model.py:94-105: frozen all-ones Conv2DTranspose k2 s2 (Q3): every
        output channel = sum over input channels, nearest 2x."""
        C = x.shape[1]
        w = torch.ones(C, C, 2, 2, dtype=x.dtype)
        return F.conv_transpose2d(x, w, stride=2)

    def feature_maps(self, x, training):
        """model.py:356-421 -> (fm1, fm2, fm3) NCHW [N, A*(5+K), G, G]."""
        self.batch_stats = []
        it = iter(range(len(self.specs)))
        raw = lambda t: self._conv_layer(t, next(it), training)
        count = [0, 0]

        def done(y):
            """y = the layer's value after the residual add; the next layer consumes its (bf16-rounded) image."""
            j = count[0]
            count[0] += 1
            if self.trace_exact is not None:
                self.trace_exact.append(y.detach())
            y = self._r(y)
            if self.trace is not None:
                self.trace.append(y.detach())
            return self.force['layers'][j].to(y.dtype) if self.force is not None else y

        def cl(t):
            i = next(it)
            y = self._conv_layer(t, i, training)
            return done(y) if self.specs[i]['bn'] else y      # heads stay fp32

        def up(t):
            y = self._upsample_2x(t)
            j = count[1]
            count[1] += 1
            if self.up_trace is not None:
                self.up_trace.append(y.detach())      # before rounding, like trace_exact
            y = self._r(y)
            return self.force['up'][j].to(y.dtype) if self.force is not None else y

        def feature_block(inp, reps):              # model.py:42-48 (Q2: adds the BLOCK input)
            layer = inp
            for _ in range(reps):
                layer = cl(layer)
                layer = done(inp + raw(layer))             # bf16 path: the residual is added before the single rounding
            return layer

        def yolo_block(inp):                        # model.py:51-59
            for _ in range(5):
                inp = cl(inp)
            route = inp
            return route, cl(inp)

        x = cl(x)
        x = cl(x)
        x = feature_block(x, 1)
        x = cl(x)
        x = feature_block(x, 2)
        x = cl(x)
        route1 = x = feature_block(x, BLOCK_COUNT)
        x = cl(x)
        route2 = x = feature_block(x, BLOCK_COUNT)
        x = cl(x)
        route3 = feature_block(x, BLOCK_COUNT // 2)

        route, x = yolo_block(route3)
        fm1 = cl(x)
        x = cl(route)
        x = torch.cat([up(x), route2], dim=1)
        route, x = yolo_block(x)
        fm2 = cl(x)
        x = cl(route)
        x = torch.cat([up(x), route1], dim=1)
        route, x = yolo_block(x)
        fm3 = cl(x)
        return fm1, fm2, fm3


def reorg_layer(fm, img_size, anchors, num_classes):
    """model.py:122-167.  fm NCHW -> (xy_offset[Gh,Gw,1,2], boxes[N,Gh,Gw,A,4]
    (cx,cy,w,h px), obj_logits[...,1], cls_logits[...,K])."""
    N, _, Gh, Gw = fm.shape
    A = len(anchors)
    dt = fm.dtype
    stride = torch.tensor([img_size[0] // Gh, img_size[1] // Gw], dtype=dt)     # (s_y, s_x) applied to (x, y): Q6
    f = fm.permute(0, 2, 3, 1).reshape(N, Gh, Gw, A, 5 + num_classes)
    box_xy, box_wh, obj, cls = torch.split(f, [2, 2, 1, num_classes], dim=-1)
    gx = torch.arange(Gw, dtype=dt)
    gy = torch.arange(Gh, dtype=dt)
    b, a = torch.meshgrid(gy, gx, indexing='ij')    # tf.meshgrid(x, y) 'xy' (App. C8): a[i,j]=x[j], b[i,j]=y[i]
    xy_offset = torch.stack([a, b], dim=-1).reshape(Gh, Gw, 1, 2)
    box_xy = (torch.sigmoid(box_xy) + xy_offset) * stride
    box_wh = torch.exp(box_wh) * torch.tensor(anchors, dtype=dt)
    return xy_offset, torch.cat([box_xy, box_wh], dim=-1), obj, cls


def decode(fms, img_size, anchors, num_classes):
    """model.py:169-212 -> [N, Nb, 5+K] rows [x0,y0,x1,y1,obj,cls...]; scale
    order coarse->fine, then row, col, anchor."""
    bl, ol, cl = [], [], []
    for fm in fms:
        _, boxes, obj, cls = reorg_layer(fm, img_size, anchors, num_classes)
        N = fm.shape[0]
        bl.append(boxes.reshape(N, -1, 4))
        ol.append(torch.sigmoid(obj.reshape(N, -1, 1)))
        cl.append(torch.sigmoid(cls.reshape(N, -1, num_classes)))
    boxes = torch.cat(bl, 1)
    cx, cy, w, h = boxes[..., 0:1], boxes[..., 1:2], boxes[..., 2:3], boxes[..., 3:4]
    return torch.cat([cx - w / 2.0, cy - h / 2.0, cx + w / 2.0, cy + h / 2.0, torch.cat(ol, 1), torch.cat(cl, 1)], dim=-1)


def _sigmoid_ce(labels, logits):
    """tf.nn.sigmoid_cross_entropy_with_logits (App. C6)."""
    return torch.clamp(logits, min=0) - logits * labels + torch.log1p(torch.exp(-torch.abs(logits)))


def loss_layer(fm, gt, img_size, anchors, num_classes):
    """model.py:230-354 -> (xy, wh, obj, class) losses, each / local batch."""
    N, _, Gh, Gw = fm.shape
    dt = fm.dtype
    stride = torch.tensor([img_size[0] // Gh, img_size[1] // Gw], dtype=dt)
    anc = torch.tensor(anchors, dtype=dt)
    B = float(N)
    xy_offset, pred_boxes, obj_logits, cls_logits = reorg_layer(fm, img_size, anchors, num_classes)
    gt = gt.to(dt)
    object_mask = gt[..., 4:5]
    pred_xy = pred_boxes[..., 0:2]
    pred_wh = pred_boxes[..., 2:4]

    # ignore mask (Q7): IoU vs origin-centred anchor-sized boxes at GT cells   model.py:250-275
    sel = object_mask[..., 0] > 0
    true_wh = (torch.ones_like(gt[..., 2:4]) * anc)[sel]            # [V,2]
    if true_wh.shape[0] == 0:
        ignore = torch.ones_like(object_mask)                       # reduce_max over empty = -inf < .5
    else:
        true_xy = torch.zeros_like(true_wh)
        pxy = pred_xy.unsqueeze(-2)
        pwh = pred_wh.unsqueeze(-2)
        imin = torch.maximum(pxy - pwh / 2.0, true_xy - true_wh / 2.0)    # broadcast_iou model.py:62-91
        imax = torch.minimum(pxy + pwh / 2.0, true_xy + true_wh / 2.0)
        iwh = torch.clamp(imax - imin, min=0.0)
        inter = iwh[..., 0] * iwh[..., 1]
        iou = inter / (pwh[..., 0] * pwh[..., 1] + true_wh[..., 0] * true_wh[..., 1] - inter)
        ignore = (iou.max(dim=-1).values < 0.5).to(dt).unsqueeze(-1)
    ignore = ignore.detach()
    valid = (object_mask + (1 - object_mask) * ignore).detach()
    object_mask = object_mask.detach()

    obj_loss = (valid * _sigmoid_ce(object_mask, obj_logits)).sum() / B             # :286-287
    cls_loss = (object_mask * _sigmoid_ce(gt[..., 5:], cls_logits)).sum() / B      # :293-294

    true_xy = gt[..., 0:2] / stride - xy_offset                                     # :313-333
    pxy = pred_boxes[..., 0:2] / stride - xy_offset
    true_xy = torch.clamp(true_xy, 0.01, 0.99)
    pxy = torch.clamp(pxy, 0.01, 0.99)
    true_xy = -torch.log(1.0 / true_xy - 1.0)
    pxy = -torch.log(1.0 / pxy - 1.0)

    true_twh = gt[..., 2:4] / anc                                                   # :337-345
    pred_twh = pred_boxes[..., 2:4] / anc
    true_twh = torch.where(true_twh == 0, torch.ones_like(true_twh), true_twh)
    pred_twh = torch.where(pred_twh == 0, torch.ones_like(pred_twh), pred_twh)
    true_twh = torch.log(torch.clamp(true_twh, 1e-9, 1e9)).detach()
    pred_twh = torch.log(torch.clamp(pred_twh, 1e-9, 1e9))
    true_xy = true_xy.detach()

    xy_loss = (torch.square(true_xy - pxy) * object_mask).sum() / B                 # :351-352
    wh_loss = (torch.square(true_twh - pred_twh) * object_mask).sum() / B
    return xy_loss, wh_loss, obj_loss, cls_loss


def compute_loss(fms, gts, img_size, anchors, num_classes):
    """model.py:214-228 -> (total, xy, wh, conf, class)."""
    xy = wh = cf = cs = 0.0
    for fm, gt in zip(fms, gts):
        a, b, c, d = loss_layer(fm, gt, img_size, anchors, num_classes)
        xy, wh, cf, cs = xy + a, wh + b, cf + c, cs + d
    return xy + wh + cf + cs, xy, wh, cf, cs


class AdamState:
    """tf.keras.optimizers.Adam (App. C5): beta1 .9, beta2 .999, eps 1e-7,
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps)."""

    def __init__(self, tensors, lr):
        self.m = [torch.zeros_like(t) for t in tensors]
        self.v = [torch.zeros_like(t) for t in tensors]
        self.t = 0
        self.lr = lr
        self.b1, self.b2, self.eps = 0.9, 0.999, 1e-7

    def lr_t(self):
        return self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)

    def step(self, tensors, grads):
        self.t += 1
        a = self.lr_t()
        with torch.no_grad():
            for p, g, m, v in zip(tensors, grads, self.m, self.v):
                m += (g - m) * (1 - self.b1)
                v += (g * g - v) * (1 - self.b2)
                p -= (m * a) / (torch.sqrt(v) + self.eps)


def train_step(net, adam, images, gts, img_size, anchors, num_classes, global_batch_size, apply=True):
    """model.py:481-508 for one replica.  Returns dict(loss, parts, grads,
    feature maps).  BN moving stats updated per App. C4."""
    for t in net.trainable():
        t.grad = None
    fms = net.feature_maps(images, training=True)
    total, xy, wh, cf, cs = compute_loss(fms, gts, img_size, anchors, num_classes)
    loss_value = total / float(global_batch_size)
    params = net.trainable()
    grads = torch.autograd.grad(loss_value, params, allow_unused=True)
    grads = [g if g is not None else torch.zeros_like(p) for g, p in zip(grads, params)]
    if apply:
        adam.step(params, grads)
        bi = 0
        with torch.no_grad():
            for sp, q in zip(net.specs, net.p):
                if not sp['bn']:
                    continue
                mean, var, cnt = net.batch_stats[bi]
                bi += 1
                q['mean'].mul_(BN_MOMENTUM).add_(mean * (1 - BN_MOMENTUM))
                q['var'].mul_(BN_MOMENTUM).add_(var * (cnt / max(cnt - 1.0, 1.0)) * (1 - BN_MOMENTUM))
    return dict(loss=float(loss_value), parts=[float(xy), float(wh), float(cf), float(cs)],
                grads=[g.detach() for g in grads], fms=[f.detach() for f in fms])


def test_step(net, images, gts, img_size, anchors, num_classes, global_batch_size):
    """model.py:517-534: BN with moving stats, loss only."""
    with torch.no_grad():
        fms = net.feature_maps(images, training=False)
        total, xy, wh, cf, cs = compute_loss(fms, gts, img_size, anchors, num_classes)
    return float(total / float(global_batch_size)), [float(xy), float(wh), float(cf), float(cs)]
