#!/opt/conda/bin/python3.9
"""Generate golden vectors by IMPORTING the reference's own NumPy code.

Run only in the build container (the reference does not travel):
    cd /tmp && PYTHONDONTWRITEBYTECODE=1 /opt/conda/bin/python3.9 \
        /root/repo/tests/golden/make_golden.py
Outputs small .npz / .json fixtures next to this script.  Nothing from the
reference is copied: the fixtures hold inputs and the reference's outputs only.

What is imported and run from /root/reference:
  bbox_utils.{compute_iou, single_class_nms, per_class_nms, filter_small_boxes,
              write_boxes_from_xywhc, write_boxes_from_ltrbpc}
  imagereader.zscore_normalize, ImageReader.__format_boxes   (pure NumPy)
  inference_tiled.{convert_image_to_tiles, inference_image_tiled} driven by a
      deterministic fake model callable (pure NumPy)
  augment.{augment_boxes, apply_affine_transformation_boxes, apply_affine_transformation,
           augment_image_box_pair} under a seeded np.random (NumPy + scikit-image + SciPy)
The modules import tensorflow / lmdb / isg_ai_pb2 at top level although the
functions above never touch them; empty placeholder modules satisfy those
imports (TensorFlow itself is NOT available: the network/loss path stays
"parity unpinned").
"""
import io
import json
import os
import sys
import types
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'

# ---- placeholder modules for imports the exercised functions never use -----
tf = types.ModuleType('tensorflow')
tf.__version__ = '2.1.0'
tf.function = lambda f: f
tf.convert_to_tensor = lambda x: x
sys.modules['tensorflow'] = tf
sys.modules['lmdb'] = types.ModuleType('lmdb')
pb = types.ModuleType('isg_ai_pb2')
pb.ImageYoloBoxesPair = type('ImageYoloBoxesPair', (), {})
sys.modules['isg_ai_pb2'] = pb
if not hasattr(np, 'bool'):
    np.bool = bool          # inference_tiled.py:236 uses the removed alias (Q13)
sys.path.insert(0, REF)

import bbox_utils           # noqa: E402
import imagereader          # noqa: E402
import inference_tiled      # noqa: E402


def synth_rows(seed, nb, K, img=416, dense=False, wh=(33, 300)):
    """SURVEY 8d 'NMS stress' rows [nb, 5+K] float32."""
    rng = np.random.default_rng(seed)
    cx = rng.uniform(0, img, nb)
    cy = rng.uniform(0, img, nb)
    w = rng.uniform(wh[0], wh[1], nb)
    h = rng.uniform(wh[0], wh[1], nb)
    obj = rng.uniform(0, 1, nb)
    if not dense:
        obj = obj ** 8
    cls = rng.uniform(0, 1, (nb, K))
    rows = np.concatenate([np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, obj], 1), cls], 1)
    rows = rows.astype(np.float32)
    # np.argsort's tie order is unspecified (App. C9): re-draw class probs of tied rows until the
    # float32 scores sqrt(obj*cls) of every class are pairwise distinct
    for _ in range(100):
        sc = np.sqrt(rows[:, 5:] * rows[:, 4:5])
        tied = np.zeros(nb, bool)
        for c in range(K):
            _, inv, cnt = np.unique(sc[:, c], return_inverse=True, return_counts=True)
            tied |= cnt[inv] > 1
        if not tied.any():
            break
        rows[tied, 5:] = rng.uniform(0, 1, (int(tied.sum()), K)).astype(np.float32)
    else:
        raise RuntimeError('could not de-tie')
    return rows


def ref_detect(rows, min_box):
    """inference.py:72-79 chain, plus the keep indices into the original rows."""
    mask = np.logical_and(rows[:, 2] - rows[:, 0] > min_box, rows[:, 3] - rows[:, 1] > min_box)
    f = bbox_utils.filter_small_boxes(rows, min_box)
    assert f.shape[0] == mask.sum()
    b, s, l = bbox_utils.per_class_nms(f[:, 0:4], f[:, 4:5], f[:, 5:])
    if b is None:
        return None
    # recover the original row index of every kept box (rows are unique by construction)
    orig = np.where(mask)[0]
    scores = np.sqrt(f[:, 5:] * f[:, 4:5])
    keep = []
    for k in range(b.shape[0]):
        c = l[k]
        cand = np.where((f[:, 0:4] == b[k]).all(1) & (scores[:, c] == s[k]))[0]
        assert cand.size == 1, 'non-unique row'
        keep.append(orig[cand[0]])
    # tie check: candidate scores of each class must be distinct
    for c in range(f.shape[1] - 5):
        sc = scores[scores[:, c] >= np.float32(0.1), c]
        assert np.unique(sc).size == sc.size, 'score tie in fixture'
    return b, s, l, np.asarray(keep, np.int32)


def g1_nms():
    cases = [('sparse416_k2', 0, 7098, 2, 416, False),
             ('dense416_k2', 1, 7098, 2, 416, True),
             ('sparse608_k3', 2, 15162, 3, 608, False),
             ('dense416_k1', 3, 7098, 1, 416, True),
             ('small_k2', 4, 300, 2, 416, True)]
    for name, seed, nb, K, img, dense in cases:
        rows = synth_rows(seed, nb, K, img, dense)
        b, s, l, keep = ref_detect(rows, 32)
        np.savez_compressed(os.path.join(HERE, 'nms_%s.npz' % name), rows=rows, min_box=np.float32(32),
                            boxes=b, scores=s, labels=l, keep=keep)
        print(name, 'kept', keep.size)
    # empty result: all scores below threshold
    rows = synth_rows(5, 500, 2)
    rows[:, 4] = 1e-4
    assert ref_detect(rows, 32) is None
    np.savez_compressed(os.path.join(HERE, 'nms_empty.npz'), rows=rows, min_box=np.float32(32))
    # single candidate
    rows = synth_rows(6, 500, 2)
    rows[:, 4] = 1e-4
    rows[123, 4] = 0.9
    rows[123, 5] = 0.8
    rows[123, 6] = 1e-4
    b, s, l, keep = ref_detect(rows, 32)
    np.savez_compressed(os.path.join(HERE, 'nms_single.npz'), rows=rows, min_box=np.float32(32), boxes=b, scores=s, labels=l, keep=keep)
    # all boxes too small
    rows = synth_rows(7, 400, 2, wh=(5, 30))
    assert ref_detect(rows, 32) is None
    np.savez_compressed(os.path.join(HERE, 'nms_allsmall.npz'), rows=rows, min_box=np.float32(32))


def g2_units():
    rng = np.random.default_rng(10)
    boxes = rng.uniform(0, 100, (64, 4)).astype(np.float32)
    boxes[:, 2:] = boxes[:, :2] + rng.uniform(1, 60, (64, 2)).astype(np.float32)
    boxes[5, 2] = boxes[5, 0]                      # zero-area box
    boxes[9] = boxes[8]                            # identical boxes, IoU exactly 1
    # a pair with IoU exactly at the threshold 0.5 in float32: [0,0,2,1] vs [0,0,1,1]
    boxes[20] = [0, 0, 2, 1]
    boxes[21] = [0, 0, 1, 1]
    scores = rng.permutation(64).astype(np.float32) / 64 + 0.001
    ious = np.stack([bbox_utils.compute_iou(boxes[i], boxes) for i in range(64)])
    keeps = {}
    for thr in (0.3, 0.5, 0.0):
        keeps['keep_%g' % thr] = np.asarray(bbox_utils.single_class_nms(boxes, scores, thr), np.int32)
    np.savez_compressed(os.path.join(HERE, 'nms_units.npz'), boxes=boxes, scores=scores, ious=ious, **keeps)


def g3_tiles():
    out = {}
    for name, (h, w, c), ts in [('4096_608', (4096, 4096, 1), (608, 608)), ('1300_608', (1300, 1300, 3), (608, 608)),
                                ('500x700_512', (500, 700, 1), (512, 512)), ('416_416', (416, 416, 3), (416, 416))]:
        rng = np.random.default_rng(20)
        img = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        tiles, xs, ys = inference_tiled.convert_image_to_tiles(img, list(ts))
        sums = [int(t.astype(np.int64).sum()) for t in tiles]
        corner = [t[0:2, 0:2, 0].astype(int).tolist() for t in tiles]
        out[name] = dict(shape=[h, w, c], tile=list(ts), x=[int(v) for v in xs], y=[int(v) for v in ys],
                         tile_shapes=[list(t.shape) for t in tiles], sums=sums, corner=corner, seed=20)
        print(name, len(tiles), 'tiles')
    with open(os.path.join(HERE, 'tiles.json'), 'w') as fh:
        json.dump(out, fh)


class FakeModel:
    """Deterministic stand-in for the network: call i returns seeded rows."""

    def __init__(self, tile, K=2, nb=400):
        self.i = 0
        self.calls = []
        self.tile, self.K, self.nb = tile, K, nb

    def __call__(self, batch, training=False):
        rows = synth_rows(1000 + self.i, self.nb, self.K, img=self.tile, dense=False, wh=(20, 120))
        # make scores livelier so that every tile keeps a few boxes
        rows[:, 4] = np.sqrt(np.sqrt(rows[:, 4]))
        self.i += 1
        self.calls.append(rows)
        return rows[None]


def g4_tiled():
    rng = np.random.default_rng(30)
    img = rng.integers(0, 256, (1500, 1900, 1), dtype=np.uint8)
    old = sys.stdout
    sys.stdout = io.StringIO()
    try:
        fm = FakeModel(608)
        pred = inference_tiled.inference_image_tiled(fm, img, [608, 608], 32)
    finally:
        sys.stdout = old
    np.savez_compressed(os.path.join(HERE, 'tiled_e2e.npz'), img_shape=np.asarray(img.shape), tile=np.asarray([608, 608]),
                        min_roi=np.int32(32), pred=pred, model_rows=np.stack(fm.calls))
    print('tiled e2e rois', pred.shape)


def g5_labels():
    out = {}
    for name, anchors, K, size in [('a2k2', [(64, 384), (384, 64)], 2, (416, 416, 3)),
                                   ('a3k3', [(32, 32), (128, 128), (256, 256)], 3, (416, 416, 1))]:
        rd = object.__new__(imagereader.ImageReader)
        rd.anchors = anchors
        rd.image_size = list(size)
        rd.number_classes = K
        rng = np.random.default_rng(40)
        n = 6
        wh = rng.integers(20, 300, (n, 2))
        xy = np.stack([rng.integers(0, size[1] - wh[:, 0]), rng.integers(0, size[0] - wh[:, 1])], 1)
        xy = np.maximum(xy, 0)
        boxes = np.concatenate([xy, wh, rng.integers(0, K, (n, 1))], 1).astype(np.int32)
        lab = rd._ImageReader__format_boxes(boxes.copy())
        empty = rd._ImageReader__format_boxes(np.zeros((0, 5), np.int32))
        assert all(e.sum() == 0 for e in empty)
        out[name + '_boxes'] = boxes
        out[name + '_anchors'] = np.asarray(anchors, np.float32)
        out[name + '_size'] = np.asarray(size)
        for i, l in enumerate(lab):
            out['%s_label%d' % (name, i + 1)] = l
    np.savez_compressed(os.path.join(HERE, 'labels.npz'), **out)


def g6_zscore():
    rng = np.random.default_rng(50)
    a = rng.integers(0, 256, (37, 41, 3), dtype=np.uint8)
    b = (rng.uniform(0, 1, (16, 16, 1)) * 0.5).astype(np.float32)      # std <= 1 branch
    c = rng.integers(0, 65535, (3, 20, 24)).astype(np.uint16)
    np.savez_compressed(os.path.join(HERE, 'zscore.npz'), a=a, a_out=imagereader.zscore_normalize(a),
                        b=b, b_out=imagereader.zscore_normalize(b), c=c, c_out=imagereader.zscore_normalize(c))


def g7_csv():
    rng = np.random.default_rng(60)
    xywhc = rng.integers(0, 400, (7, 5)).astype(np.int32)
    ltrbpc = np.concatenate([rng.integers(0, 400, (7, 4)).astype(np.float64), rng.uniform(0, 1, (7, 1)), rng.integers(0, 2, (7, 1)).astype(np.float64)], 1)
    ltrbpc[:, 2:4] += ltrbpc[:, 0:2]
    with tempfile.TemporaryDirectory() as d:
        bbox_utils.write_boxes_from_xywhc(xywhc, os.path.join(d, 'a.csv'))
        bbox_utils.write_boxes_from_ltrbpc(ltrbpc, os.path.join(d, 'b.csv'))
        ta = open(os.path.join(d, 'a.csv')).read()
        tb = open(os.path.join(d, 'b.csv')).read()
    with open(os.path.join(HERE, 'csv.json'), 'w') as fh:
        json.dump(dict(xywhc=xywhc.tolist(), xywhc_text=ta, ltrbpc=ltrbpc.tolist(), ltrbpc_text=tb), fh)


def g8_augment():
    """augment.py: box jitter, box / image affine helpers with explicit parameters, and the full pair augmentation
    under a seeded global NumPy RNG (the reference draws from np.random directly)."""
    import augment
    out = {}
    rng = np.random.default_rng(8)
    boxes = np.stack([rng.integers(0, 300, 12), rng.integers(0, 200, 12), rng.integers(20, 120, 12), rng.integers(20, 100, 12),
                      rng.integers(0, 3, 12)], 1).astype(np.int32)
    out['boxes'] = boxes
    np.random.seed(123)
    out['jitter'] = augment.augment_boxes(boxes.copy(), 0.05, 0.08, (260, 420))
    cases = []
    for i, (rx, ry, sx, sy, dx, dy, crop) in enumerate([(0, 0, 1.0, 1.0, 0, 0, (260, 420)), (1, 0, 1.0, 1.0, 30, 10, (200, 300)),
                                                        (0, 1, 1.1, 0.9, 25, 5, (180, 320)), (1, 1, 0.8, 1.2, 0, 40, (160, 200)),
                                                        (0, 0, 1.0, 1.0, 400, 300, (64, 64))]):
        r = augment.apply_affine_transformation_boxes(boxes.copy(), crop, rx, ry, sx, sy, dx, dy)
        out['affine_boxes_%d' % i] = np.zeros((0, 5), np.int32) if r is None else r
        cases.append([rx, ry, sx, sy, dx, dy, list(crop)])
    out['affine_cases'] = np.asarray([c[:6] for c in cases], np.float64)
    out['affine_crops'] = np.asarray([c[6] for c in cases], np.int32)
    img = rng.integers(0, 256, (90, 120, 3)).astype(np.float32)
    img2 = rng.integers(0, 256, (70, 100)).astype(np.float32)
    out['img'], out['img2'] = img, img2
    icases = [(0, 0, 1.0, 1.0, (64, 96)), (1, 1, 1.0, 1.0, (90, 120)), (1, 0, 1.15, 0.9, (64, 96)), (0, 1, 0.85, 1.3, (56, 80))]
    for i, (rx, ry, sx, sy, crop) in enumerate(icases):
        np.random.seed(50 + i)
        I, dx, dy = augment.apply_affine_transformation(img, rx, ry, sx, sy, crop)
        out['affine_img_%d' % i] = np.asarray(I, np.float32)
        out['affine_img_%d_dxdy' % i] = np.asarray([dx, dy])
    np.random.seed(60)
    I, dx, dy = augment.apply_affine_transformation(img2, 1, 0, 1.2, 0.95, (60, 90))
    out['affine_img2'] = np.asarray(I, np.float32)
    out['affine_img2_dxdy'] = np.asarray([dx, dy])
    out['affine_img_cases'] = np.asarray([c[:4] for c in icases], np.float64)
    out['affine_img_crops'] = np.asarray([c[4] for c in icases], np.int32)
    # the whole pair augmentation: no rescale (bit-comparable pixels) and with rescale
    pair_boxes = np.asarray([[10, 12, 40, 30, 0], [60, 20, 35, 50, 1], [80, 50, 30, 30, 2]], np.int32)
    out['pair_boxes'] = pair_boxes
    for i, kw in enumerate([dict(reflection_flag=True, crop_to=(64, 96), noise_augmentation_severity=0.02, scale_augmentation_severity=0,
                                 blur_augmentation_max_sigma=2, box_size_augmentation_severity=0.03, box_location_jitter_severity=0.03),
                            dict(reflection_flag=True, crop_to=(64, 96), noise_augmentation_severity=0.02, scale_augmentation_severity=0.1,
                                 blur_augmentation_max_sigma=2, box_size_augmentation_severity=0.03, box_location_jitter_severity=0.03)]):
        for seed in (1, 2, 3):
            np.random.seed(seed)
            I, b = augment.augment_image_box_pair(img.copy(), pair_boxes.copy(), **kw)
            out['pair_%d_%d_img' % (i, seed)] = I
            out['pair_%d_%d_boxes' % (i, seed)] = np.zeros((0, 5), np.int32) if b is None else b
    np.savez_compressed(os.path.join(HERE, 'augment.npz'), **out)
    print('augment: %d arrays' % len(out))


def g9_misc():
    """bbox_utils.write_boxes_from_ltrbc text and imagereader.inverse_format_boxes on a label tensor produced by the
    reference's own __format_boxes."""
    rng = np.random.default_rng(9)
    ltrbc = np.stack([rng.integers(0, 100, 6), rng.integers(0, 100, 6), rng.integers(100, 300, 6), rng.integers(100, 300, 6),
                      rng.integers(0, 3, 6)], 1).astype(np.int32)
    with tempfile.TemporaryDirectory() as d:
        fp = os.path.join(d, 'a.csv')
        bbox_utils.write_boxes_from_ltrbc(ltrbc, fp)
        text = open(fp).read()
    rd = object.__new__(imagereader.ImageReader)
    rd.anchors = [(64, 384), (384, 64)]
    rd.image_size = [416, 416, 3]
    rd.number_classes = 2
    boxes = np.asarray([[20, 30, 90, 60, 0], [200, 220, 120, 150, 1], [300, 40, 50, 200, 1]], np.int32)
    l1, l2, l3 = rd._ImageReader__format_boxes(boxes.copy())
    label = np.stack([l3, l3])          # batch of 2
    inv = imagereader.inverse_format_boxes(label.copy(), 1)
    with open(os.path.join(HERE, 'misc.json'), 'w') as fh:
        json.dump(dict(ltrbc=ltrbc.tolist(), ltrbc_text=text, fmt_boxes=boxes.tolist(), label3_nonzero=np.argwhere(l3[..., 4] > 0).tolist(), inverse=np.asarray(inv).tolist()), fh)


if __name__ == '__main__':
    g1_nms()
    g2_units()
    g3_tiles()
    g4_tiled()
    g5_labels()
    g6_zscore()
    g7_csv()
    g8_augment()
    g9_misc()
    print('done')
