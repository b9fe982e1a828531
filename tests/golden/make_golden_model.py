#!/usr/bin/env python3
"""Execute the TEXT of the reference's model.py over a recording / NumPy stand-in of the TensorFlow calls it makes.

Run only in the build container (the reference does not travel):
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tests/golden/make_golden_model.py

TensorFlow is not installed here, so this does NOT pin parity with TensorFlow's arithmetic (VERDICT r2 item 8 says so too:
"stand-in op semantics pin nothing").  What it removes is the failure class "the builder misread model.py": the
reference's own ``YoloV3.__init__`` -> ``build_feature_maps`` -> ``convert_feature_map_to_inference_detections`` and
``reorg_layer`` / ``loss_layer`` / ``compute_loss`` are executed as written, and

  arch.json      <- every Keras layer call the reference makes while it builds the graph, in call order: kind, filters,
                    kernel, stride, padding, activation, trainable, initializer, name, input / output shapes, the ids of the
                    tensors it reads, plus every tf.add / tf.concat with its operands.  Pure recording: no arithmetic.
  model_fwd.npz  <- reorg_layer, convert_feature_map_to_inference_detections and loss_layer / compute_loss evaluated on
                    seeded feature maps and labels (13 / 26 / 52 grids, V = 0 and V > 0 ground-truth boxes) with the tf.*
                    element-wise calls restated in NumPy float32 (tf.nn.sigmoid_cross_entropy_with_logits by its documented
                    formula max(x,0) - x*z + log1p(exp(-|x|)); reduce_max over an empty V axis = -inf).

tests/test_cpu_oracle.py compares oracle/model.py and the product's layer table with these on the CPU.
"""
import json
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'

# ---------------------------------------------------------------------------------------------------------------
# the stand-in: a tensor wrapper with TF's .shape.as_list(), NumPy float32 arithmetic underneath
# ---------------------------------------------------------------------------------------------------------------
class Shape(tuple):
    def as_list(self):
        return list(self)


_next_id = [0]


class T:
    __array_priority__ = 1000

    def __init__(self, v):
        self.v = np.asarray(v)
        self.id = _next_id[0]
        _next_id[0] += 1

    @property
    def shape(self):
        return Shape(int(d) for d in self.v.shape)

    @property
    def dtype(self):
        return self.v.dtype

    def __getitem__(self, k):
        return T(self.v[k])

    def __len__(self):
        return len(self.v)

    def __float__(self):
        return float(self.v)

    def _b(self, o, f):
        return T(f(self.v, un(o)))

    def __add__(self, o): return self._b(o, np.add)
    def __radd__(self, o): return T(np.add(un(o), self.v))
    def __sub__(self, o): return self._b(o, np.subtract)
    def __rsub__(self, o): return T(np.subtract(un(o), self.v))
    def __mul__(self, o): return self._b(o, np.multiply)
    def __rmul__(self, o): return T(np.multiply(un(o), self.v))
    def __truediv__(self, o): return self._b(o, np.divide)
    def __rtruediv__(self, o): return T(np.divide(un(o), self.v))
    def __floordiv__(self, o): return self._b(o, np.floor_divide)
    def __lt__(self, o): return self._b(o, np.less)
    def __neg__(self): return T(-self.v)


def un(x):
    """TF converts python scalars / lists to the other operand's dtype (float32 here); never promote to float64."""
    if isinstance(x, T):
        return x.v
    a = np.asarray(x)
    if a.dtype == np.float64:
        a = a.astype(np.float32)
    return a


def wrap(f):
    return lambda *a, **k: T(f(*[un(x) for x in a], **k))


GRAPH = []          # recorded layer / op calls


def rec(kind, out, inputs, **attrs):
    GRAPH.append(dict(kind=kind, out=out.id, out_shape=list(out.shape), inputs=[t.id for t in inputs], **attrs))
    return out


tf = types.ModuleType('tensorflow')
tf.__version__ = '2.1.0'
tf.float32, tf.int32 = np.float32, np.int32
tf.function = lambda f: f
tf.convert_to_tensor = lambda x: x if isinstance(x, T) else T(x)
tf.cast = lambda x, dt: T(un(x).astype({'bool': np.bool_}.get(dt, dt)))
tf.transpose = lambda x, perm: T(np.transpose(un(x), perm))
tf.reshape = lambda x, s: T(np.reshape(un(x), [int(d) for d in s]))
tf.split = lambda x, sizes, axis: [T(p) for p in np.split(un(x), np.cumsum(sizes)[:-1], axis=axis)]
tf.range = lambda n, dtype=np.int32: T(np.arange(n, dtype=dtype))
tf.meshgrid = lambda a, b: [T(m) for m in np.meshgrid(un(a), un(b))]        # default indexing 'xy', as tf.meshgrid
tf.expand_dims = lambda x, ax: T(np.expand_dims(un(x), ax))
tf.maximum, tf.minimum, tf.exp, tf.square = wrap(np.maximum), wrap(np.minimum), wrap(np.exp), wrap(np.square)
tf.zeros_like, tf.ones_like = wrap(np.zeros_like), wrap(np.ones_like)
tf.boolean_mask = lambda x, m: T(un(x)[un(m)])
tf.reduce_max = lambda x, axis: T(np.max(un(x), axis=axis, initial=-np.inf))
tf.reduce_sum = lambda x: T(np.sum(un(x), dtype=np.float32))
tf.stop_gradient = lambda x: x
tf.shape = lambda x: T(np.asarray(un(x).shape, np.int32))
tf.clip_by_value = lambda x, lo, hi: T(np.clip(un(x), np.float32(lo), np.float32(hi)))
tf.constant = lambda v, dtype=np.float32: T(np.asarray(v, dtype))
tf.where = lambda condition, x, y: T(np.where(un(condition), un(x), un(y)))
tf.equal = lambda a, b: T(un(a) == un(b))
tf.ones_initializer = lambda: 'ones'


def _sigmoid(x):
    x = un(x)
    return T((1.0 / (1.0 + np.exp(-x.astype(np.float32)))).astype(np.float32))


tf.sigmoid = _sigmoid
tf.nn = types.SimpleNamespace(
    sigmoid=_sigmoid, leaky_relu='leaky_relu',
    sigmoid_cross_entropy_with_logits=lambda labels, logits: T(
        np.maximum(un(logits), 0) - un(logits) * un(labels) + np.log1p(np.exp(-np.abs(un(logits))))))
tf.math = types.SimpleNamespace(log=wrap(np.log))


def _add(a, b):
    return rec('add', T(un(a) + un(b)), [a, b])


def _concat(xs, axis):
    out = T(np.concatenate([un(x) for x in xs], axis=axis))
    if all(isinstance(x, T) for x in xs) and out.v.ndim == 4 and axis == 1:      # the two route concats (NCHW channel axis)
        rec('concat', out, list(xs), axis=axis)
    return out


tf.add, tf.concat = _add, _concat


def _same(size, s):
    return -(-size // s)


class Conv2D:
    def __init__(self, filters, kernel_size, padding, activation, strides, data_format, kernel_regularizer=None, name=None):
        self.a = dict(filters=int(filters), kernel=int(kernel_size), padding=padding, activation=activation or 'linear', stride=int(strides),
                      data_format=data_format, l2=kernel_regularizer, name=name, trainable=True, use_bias=True)

    def __call__(self, x):
        n, c, h, w = x.shape
        out = T(np.zeros((n, self.a['filters'], _same(h, self.a['stride']), _same(w, self.a['stride'])), np.float32))
        return rec('Conv2D', out, [x], cin=c, **self.a)


class BatchNormalization:
    def __init__(self, axis):
        self.axis = axis

    def __call__(self, x):
        return rec('BatchNormalization', T(np.zeros(x.shape, np.float32)), [x], axis=self.axis)


class Conv2DTranspose:
    def __init__(self, filters, kernel_size, padding, strides, activation, data_format, kernel_initializer, trainable):
        self.a = dict(filters=int(filters), kernel=int(kernel_size), padding=padding, stride=int(strides), activation=activation or 'linear',
                      data_format=data_format, kernel_initializer=kernel_initializer, trainable=bool(trainable))

    def __call__(self, x):
        n, c, h, w = x.shape
        return rec('Conv2DTranspose', T(np.zeros((n, self.a['filters'], h * self.a['stride'], w * self.a['stride']), np.float32)), [x], cin=c, **self.a)


MODELS = []
tf.keras = types.SimpleNamespace(
    layers=types.SimpleNamespace(Conv2D=Conv2D, BatchNormalization=BatchNormalization, Conv2DTranspose=Conv2DTranspose),
    regularizers=types.SimpleNamespace(l2=lambda l: float(l)),
    Input=lambda shape: rec('Input', T(np.zeros((1,) + tuple(shape), np.float32)), [], declared_shape=list(shape)),
    Model=lambda inputs, outputs, name: MODELS.append(dict(name=name, inputs=inputs.id, outputs=[o.id for o in (outputs if isinstance(outputs, (tuple, list)) else [outputs])])) or name,
    optimizers=types.SimpleNamespace(Adam=lambda learning_rate: dict(optimizer='Adam', learning_rate=learning_rate)),
)
sys.modules['tensorflow'] = tf
sys.path.insert(0, REF)
import model as ref_model          # noqa: E402   (/root/reference/model.py, executed as written)


def main():
    out = {}
    # ---- architecture: two configurations (benchmark: 2 anchors x 2 classes RGB; default anchors, 3 classes, grayscale) ----
    for tag, (img, k, anchors) in {'rgb416_a2_k2': ([416, 416, 3], 2, [(64, 384), (384, 64)]), 'gray96x160_a3_k3': ([96, 160, 1], 3, None)}.items():
        GRAPH.clear()
        MODELS.clear()
        y = ref_model.YoloV3(8, img, k, anchors, 1e-4)
        out[tag] = dict(img_size=img, number_classes=k, anchors=[list(a) for a in y.anchors], graph=list(GRAPH), models=list(MODELS),
                        optimizer=y.optimizer, output_shape=[float(v) for v in y.output_shape],
                        constants=dict(BLOCK_COUNT=y.BLOCK_COUNT, FILTER_COUNT=y.FILTER_COUNT, KERNEL_SIZE=y.KERNEL_SIZE,
                                       NETWORK_DOWNSAMPLE_FACTOR=y.NETWORK_DOWNSAMPLE_FACTOR, WEIGHT_DECAY=y.WEIGHT_DECAY,
                                       score_threshold=y.score_threshold, iou_threshold=y.iou_threshold))
    with open(os.path.join(HERE, 'arch.json'), 'w') as fh:
        json.dump(out, fh, separators=(',', ':'))

    # ---- decode / loss forward values --------------------------------------------------------------------------------------
    fx = {}
    rng = np.random.default_rng(20)
    for tag, (img, k, anchors, n) in {'sq': ([416, 416, 3], 2, [(64, 384), (384, 64)], 2), 'rect': ([96, 160, 1], 3, [(32, 32), (128, 128), (256, 256)], 3)}.items():
        GRAPH.clear()
        y = ref_model.YoloV3(n, img, k, anchors, 1e-4)
        A, D = len(anchors), len(anchors) * (5 + k)
        fms, gts = [], []
        for s in (32, 16, 8):
            gh, gw = img[0] // s, img[1] // s
            fms.append(rng.normal(0, 1.5, (n, D, gh, gw)).astype(np.float32))
            gt = np.zeros((n, gh, gw, A, 5 + k), np.float32)
            for b in range(n):
                if tag == 'sq' and b == 1 and s != 16:
                    continue                    # image 1 has no boxes at the 13 and 52 grids; the whole batch at scale 'v0' below has none
                for _ in range(3):
                    cy, cx, a = rng.integers(0, gh), rng.integers(0, gw), rng.integers(0, A)
                    w, h = rng.uniform(20, 300, 2)
                    gt[b, cy, cx, a, 0:4] = [(cx + rng.uniform(0.05, 0.95)) * (img[1] / gw), (cy + rng.uniform(0.05, 0.95)) * (img[0] / gh), w, h]
                    gt[b, cy, cx, a, 4] = 1.0
                    gt[b, cy, cx, a, 5:] = 0.0
                    gt[b, cy, cx, a, 5 + rng.integers(0, k)] = 1.0
            gts.append(gt)
        rows = y.convert_feature_map_to_inference_detections([T(f) for f in fms])
        fx[tag + '_rows'] = rows.v
        for i, (f, g) in enumerate(zip(fms, gts)):
            xy_offset, boxes, obj, cls = y.reorg_layer(T(f))
            fx['%s_fm%d' % (tag, i)] = f
            fx['%s_gt%d' % (tag, i)] = g
            fx['%s_reorg%d_xy_offset' % (tag, i)] = xy_offset.v
            fx['%s_reorg%d_boxes' % (tag, i)] = boxes.v
            fx['%s_loss%d' % (tag, i)] = np.asarray([float(t) for t in y.loss_layer(T(f), T(g))], np.float64)
            fx['%s_loss%d_v0' % (tag, i)] = np.asarray([float(t) for t in y.loss_layer(T(f), T(np.zeros_like(g)))], np.float64)   # V = 0
        fx[tag + '_compute_loss'] = np.asarray([float(t) for t in y.compute_loss([T(f) for f in fms], [T(g) for g in gts])], np.float64)
        fx[tag + '_meta'] = np.asarray(img + [k, n], np.int64)
        fx[tag + '_anchors'] = np.asarray(anchors, np.float32)
    np.savez_compressed(os.path.join(HERE, 'model_fwd.npz'), **fx)
    print('wrote arch.json (%d + %d graph nodes) and model_fwd.npz (%d arrays)' % (len(out['rgb416_a2_k2']['graph']), len(out['gray96x160_a3_k3']['graph']), len(fx)))


if __name__ == '__main__':
    main()
