"""GPU parity at the sizes BASELINE.json quotes (the sizes bench.py times), for the cases that the per-kernel and
small-model tests cannot stand in for because the launch planner picks other tiles / split-K factors / stream overlaps
there:

* configs[2]: the two-stream, host-launched training step at batch 8, 416x416 must equal the single-stream HIP-graph
  step bit for bit (hazards between the streams only last long enough to matter at this size);
* configs[4]: inference_tiled on a 4096x4096 image cut into 100 tiles of 608x608 with the real network on the bf16 conv
  path, against the fp32 path of the same network (rows, then detections) + properties of the output;
* configs[0]: the inference.py CLI at 416x416, csv files against the in-process pipeline and against the CPU oracle
  (box IoU, the metric BASELINE.json names).
(configs[1] and the full-size train step against the oracle live in test_gpu_model.py: the (416, 8) parameters.)
"""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'object-detection-yolov3_amd')
ANCHORS = [(64, 384), (384, 64)]
K = 2


def _iou_matrix(a, b):
    x0 = np.maximum(a[:, None, 0], b[None, :, 0])
    y0 = np.maximum(a[:, None, 1], b[None, :, 1])
    x1 = np.minimum(a[:, None, 2], b[None, :, 2])
    y1 = np.minimum(a[:, None, 3], b[None, :, 3])
    inter = np.maximum(x1 - x0, 0) * np.maximum(y1 - y0, 0)
    aa = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    ab = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    return inter / np.maximum(aa[:, None] + ab[None, :] - inter, 1e-9)


def matched_fraction(a_boxes, a_cls, b_boxes, b_cls, iou=0.8):
    """Share of the boxes in a that have a box of the same class in b with IoU >= iou."""
    if len(a_boxes) == 0:
        return 1.0
    if len(b_boxes) == 0:
        return 0.0
    m = _iou_matrix(np.asarray(a_boxes, np.float64), np.asarray(b_boxes, np.float64))
    m = np.where(np.asarray(a_cls)[:, None] == np.asarray(b_cls)[None, :], m, 0.0)
    return float((m.max(1) >= iou).mean())


def sparse_detector(yolo, images, frac=0.004, min_size=32):
    """Random-init weights give a flat score field (every box scores the same to ~1 %, so which boxes clear a threshold and
    in which order NMS visits them is rounding noise).  Make the random network a usable detector instead: (1) scale the
    three head layers so that the objectness logits of `images` have a standard deviation of 6 (saturating, bimodal scores like a trained detector, box
    sizes spread widely around the anchors); (2) shift the objectness bias so that the fraction `frac` of the
    boxes clears the 0.1 score threshold (score = sqrt(objectness * class probability), bbox_utils.py:244) AND the 32-pixel
    size filter -- found by bisection on the fp32 rows.  Returns the weights."""
    from oracle import model as om
    params = om.init_params(yolo.img_size[2], yolo.number_anchors, K, seed=17, randomize_bn=True)
    for p in params:
        if 'gamma' not in p:
            p['W'] *= 0.02
    yolo.set_weights(params)

    def logits(rows):
        obj = rows[..., 4].flatten().double().clamp(1e-12, 1 - 1e-12)
        return torch.log(obj) - torch.log1p(-obj)

    sigma = float(logits(yolo.predict(images, precision='fp32')).std())
    D = 5 + K
    for p, sp in zip(params, yolo.specs):
        if not sp.bn:
            for a in range(yolo.number_anchors):         # objectness channels only: box geometry keeps its O(1) logits
                p['W'][..., a * D + 4] *= 6.0 / max(sigma, 1e-6)
                p['b'][a * D + 4] *= 6.0 / max(sigma, 1e-6)
    yolo.set_weights(params)
    rows = yolo.predict(images, precision='fp32').double()
    logit = logits(rows)
    cls = rows[..., 5:].max(-1).values.flatten()
    big = (((rows[..., 2] - rows[..., 0]) > min_size + 1) & ((rows[..., 3] - rows[..., 1]) > min_size + 1)).flatten()
    lo, hi = -40.0, 40.0
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        share = float(((torch.sqrt(torch.sigmoid(logit + mid) * cls) >= 0.1) & big).double().mean())
        lo, hi = (lo, mid) if share > frac else (mid, hi)
    shift = 0.5 * (lo + hi)
    for p, sp in zip(params, yolo.specs):
        if not sp.bn:
            for a in range(yolo.number_anchors):
                p['b'][a * D + 4] += shift
    yolo.set_weights(params)
    rows = yolo.predict(images, precision='fp32')
    sc = torch.sqrt(rows[..., 4:5] * rows[..., 5:]).max(-1).values
    big = ((rows[..., 2] - rows[..., 0]) > min_size) & ((rows[..., 3] - rows[..., 1]) > min_size)
    print('sparse_detector: head scale %.1f, shift %.3f: %d boxes clear the score threshold and the size filter (%.4f), score max %.3f' % (
        6.0 / max(sigma, 1e-6), shift, int(((sc >= 0.1) & big).sum()), float(((sc >= 0.1) & big).double().mean()), float(sc.max())))
    return params


def test_two_stream_step_equals_graph_step_at_benchmark_size():
    """Batch 8, 416x416 (what bench.py times).  Model a launches from the host with every kernel gradient on a second
    stream (double-buffered dz, one shared slab workspace in stream order); model b replays the same kernels as a
    single-stream HIP graph.  Same kernels, same accumulation orders: gradients, weights and moving statistics must agree
    BIT FOR BIT after every step, and so must a re-run of the same step (run-to-run determinism)."""
    import bench
    from yolo3.model import YoloV3
    a = YoloV3(8, [416, 416, 3], K, ANCHORS, learning_rate=1e-4, seed=1)
    b = YoloV3(8, [416, 416, 3], K, ANCHORS, learning_rate=1e-4, seed=1, use_graph=True)
    c = YoloV3(8, [416, 416, 3], K, ANCHORS, learning_rate=1e-4, seed=1)
    assert a._plan(8, True).side is not None and b._plan(8, True).side is None
    g = torch.Generator().manual_seed(100)
    for s in range(3):
        images = torch.randn(8, 3, 416, 416, generator=g).cuda()
        gts = [torch.from_numpy(x).cuda() for x in bench.synth_labels(np.random.default_rng(3 + s), 8)]
        la, lb, lc = (float(m.train_step((images, gts))) for m in (a, b, c))
        torch.cuda.synchronize()
        assert np.isfinite(la) and la == lb == lc, (s, la, lb, lc)
        for m in (b, c):
            assert torch.equal(a.grads, m.grads), 'gradients differ at step %d' % s
            assert torch.equal(a.params, m.params) and torch.equal(a.moving, m.moving), 'weights / moving stats differ at step %d' % s
            assert torch.equal(a.adam_m, m.adam_m) and torch.equal(a.adam_v, m.adam_v)


def test_tiled_fp32_pipeline_against_the_oracle_chain():
    """BASELINE.json configs[4] against the ORACLE, end to end (the 4k test below compares two HIP paths with each other): a
    1000 x 900 x 3 image, 25 tiles of 416 x 416 (96-px ghost border).  HIP: inference_image_tiled (banded upload, fused tile
    preprocessing, fp32 network, decode, NMS on the GPU, ghost-band merge).  Oracle: the host tiler (pinned by the reference's
    tiles.json) -> NumPy z-score per tile -> oracle network fp32 -> decode -> oracle NMS -> the same merge / finalize code
    (pinned by tiled_e2e.npz; inference_tiled.py:185-310).  Boxes must match at IoU >= 0.8, class for class, both ways.
    (The bf16 conv path is NOT compared at the level of merged detections, and no code here pretends to: measured in round 3
    against the oracle with the same rounding points (Net.bf16), a random network gave 0.54 / 0.50 of all and 0.65 / 0.62 of the
    boxes with score >= 0.13 -- the distance of the bf16 kernels to their emulation equals the distance of bf16 to fp32 itself
    (test_bf16_inference_matches_bf16_oracle), and greedy NMS over a random network's near-tied, overlapping candidates turns a
    2e-2 score difference into a different survivor.  The sharp statements for config 4 are test_bf16_layers_teacher_forced[608-2]
    per layer (half a bf16 ulp) and test_tiled_4k_bf16_against_fp32 for the pipeline.)"""
    import contextlib
    import io
    import inference_tiled
    from oracle import model as om
    from oracle import nms as onms
    from yolo3 import imagereader
    from yolo3.model import YoloV3
    tile, min_roi = [416, 416], 32
    rng = np.random.default_rng(21)
    yy, xx = np.mgrid[0:1000, 0:900]
    img = rng.integers(0, 256, (1000, 900, 3)) * 0.5 + 64 + 60 * np.sin(xx / 23.0)[..., None] * np.cos(yy / 31.0)[..., None]
    img = np.clip(img, 0, 255).astype(np.uint8)
    tiles, xs, ys = inference_tiled.convert_image_to_tiles(img, tile)
    assert len(tiles) == 25
    yolo = YoloV3(len(tiles), [416, 416, 3], K, ANCHORS, seed=1)
    x = torch.from_numpy(np.stack([t.astype(np.float32).transpose(2, 0, 1) for t in tiles])).cuda()
    params = sparse_detector(yolo, imagereader.zscore_normalize_device(x), frac=0.03, min_size=min_roi)
    with contextlib.redirect_stdout(io.StringIO()):
        got = inference_tiled.inference_image_tiled(yolo.get_keras_model(), img, tile, min_roi, batch_size=25)
    net = om.Net(params, 3, len(ANCHORS), K, dtype=torch.float32)
    bl, sl, cl = [], [], []
    for t, tx, ty in zip(tiles, xs, ys):
        im = t.astype(np.float32)
        sd = im.std()
        z = (im - im.mean()) / (sd if sd > 1.0 else 1.0)
        with torch.no_grad():
            fms = net.feature_maps(torch.from_numpy(z.transpose(2, 0, 1)[None].astype(np.float32)), training=False)
            orow = om.decode(fms, (416, 416, 3), ANCHORS, K).numpy()[0]
        keep = onms.detect_rows(orow, min_roi)
        if sum(len(k) for k in keep) == 0:
            continue
        boxes = np.concatenate([orow[k, 0:4] for k in keep])
        scores = np.concatenate([np.sqrt(orow[k, 5 + c] * orow[k, 4]) for c, k in enumerate(keep)])
        labels = np.concatenate([np.full(len(k), c, np.int32) for c, k in enumerate(keep)])
        r = inference_tiled.merge_tile_detections(boxes, scores, labels, tx, ty, tile, img.shape)
        if r is not None:
            bl.append(r[0]); sl.append(r[1]); cl.append(r[2])
    want = inference_tiled.finalize_predictions(bl, sl, cl, img.shape)
    assert got.shape[1] == 6 and want.shape[1] == 6 and len(want) >= 15, (got.shape, want.shape)
    fa = matched_fraction(got[:, 0:4], got[:, 5], want[:, 0:4], want[:, 5])
    fb = matched_fraction(want[:, 0:4], want[:, 5], got[:, 0:4], got[:, 5])
    print('tiled fp32 vs oracle chain: %d / %d boxes, matched %.3f / %.3f' % (len(got), len(want), fa, fb))
    assert fa >= 0.95 and fb >= 0.95, (fa, fb)


def test_tiled_4k_bf16_against_fp32():
    """BASELINE.json configs[4]: 4096 x 4096 x 3 uint8 image, 100 tiles of 608 x 608 (96-px ghost border), the real
    network in launches of 25 tiles on two streams, bf16 conv path + fp32 heads / decode / NMS.
    (1) rows of one 25-tile batch: bf16 vs fp32 network on the same z-scored tiles, relative L2;
    (2) the merged detections of the whole image: bf16 vs fp32, share of boxes matched at IoU >= 0.8 (same class);
    (3) properties: inside the image, larger than min_roi, score >= 0.1, class in range, repeatable bit for bit,
        and the number of tiles the model saw."""
    import contextlib
    import io
    import inference_tiled
    from yolo3 import imagereader
    from yolo3.model import YoloV3
    # small anchors and a 16-pixel size filter for this test: with train.py's 64 x 384 anchors a 608 tile holds only a
    # handful of boxes that survive NMS and the ghost band, too few to state an agreement on
    tile, min_roi, anchors = [608, 608], 16, [(40, 40), (64, 48)]
    big = np.random.default_rng(4).integers(0, 256, (4096, 4096, 3), dtype=np.uint8)
    # smooth blobs so that tiles differ in their statistics (per-tile z-score, Q12)
    yy, xx = np.mgrid[0:4096, 0:4096]
    big = np.clip(big * 0.5 + 60 * np.sin(xx / 97.0)[..., None] + 60 * np.cos(yy / 131.0)[..., None] + 64, 0, 255).astype(np.uint8)
    y = YoloV3(25, [608, 608, 3], K, anchors, seed=1, use_graph=True)
    table, xs, ys = inference_tiled.tile_table(4096, 4096, tile)
    assert len(xs) == 100
    img_dev = torch.from_numpy(big).cuda()
    x = inference_tiled.tiles_to_device(img_dev, 0, big.shape, torch.from_numpy(table).cuda(), 0, 25, tile)
    x = imagereader.zscore_normalize_device(x)
    # (1) on the plain random network (Glorot kernels x 0.02, randomised BatchNorm statistics: O(1) head logits)
    from oracle import model as om
    plain = om.init_params(3, len(anchors), K, seed=17, randomize_bn=True)
    for q in plain:
        if 'gamma' not in q:
            q['W'] *= 0.02
    y.set_weights(plain)
    r32 = y.predict(x, precision='fp32').clone()
    r16 = y.predict(x, precision='bf16').clone()
    assert torch.isfinite(r16).all()
    d = float((r16 - r32).double().norm() / r32.double().norm())
    dobj = float((r16[..., 4:] - r32[..., 4:]).abs().max())
    print('rows bf16 vs fp32: rel L2 %.3e, max |d score| %.3e' % (d, dobj))
    assert d <= 3e-2 and dobj <= 0.05          # 75 layers of bf16 storage (8 significant bits): a few 1e-2 end to end
    # (2), (3) on the same network turned into a sparse detector with saturating scores (heads scaled up: the bf16 error of
    # the logits scales with them, the boxes that sit AT the score cut become few)
    sparse_detector(y, x, min_size=min_roi)
    r32 = y.predict(x, precision='fp32').clone()
    r16 = y.predict(x, precision='bf16').clone()

    class Counting:
        supports_slots = True

        def __init__(self, m):
            self.m, self.tiles = m, 0

        def __call__(self, batch, training=False, slot=0):
            assert batch.shape[1:] == (3, 608, 608) and batch.is_cuda
            self.tiles += batch.shape[0]
            return self.m(batch, training=training, slot=slot)

    preds = {}
    for prec in ('fp32', 'bf16', 'bf16'):
        y.inference_precision = prec
        cm = Counting(y.get_keras_model())
        with contextlib.redirect_stdout(io.StringIO()):
            p = inference_tiled.inference_image_tiled(cm, big, tile, min_roi, batch_size=25)
        assert cm.tiles == 100
        if prec in preds:
            assert np.array_equal(preds[prec], p), 'tiled bf16 inference is not repeatable'
        preds[prec] = p
    p32, p16 = preds['fp32'], preds['bf16']
    # the batches the bf16 path plans for itself (45 + 45 + 10 tiles for this image: whole rounds of 256 workgroups on the
    # 256 x 256-tile kernel) must give the detections of the 4 x 25 split: a tile's result does not depend on its batch
    # ... and so must the fused preprocessing (run_tiles: gather + z-score + input layout in two passes over the image), which
    # the plain callable above does not offer: the runs above took the three-step host path
    class CountingFused(Counting):
        def run_tiles(self, img_dev, code, shape, table_ptr, count, tile_size=None, slot=0):
            self.tiles += count
            return self.m.run_tiles(img_dev, code, shape, table_ptr, count, tile_size=tile_size, slot=slot)

    cm = CountingFused(y.get_keras_model())
    with contextlib.redirect_stdout(io.StringIO()):
        p_auto = inference_tiled.inference_image_tiled(cm, big, tile, min_roi)
    assert cm.tiles == 100 and inference_tiled.plan_tile_batches(100, tile) == [45, 45, 10]
    assert np.array_equal(p_auto, p16), 'planned tile batches / fused preprocessing changed the detections'
    print('merged detections of the whole image: fp32 %d, bf16 %d' % (len(p32), len(p16)))
    for p in (p32, p16):
        assert p.dtype == np.float64 and p.shape[1] == 6
        assert (p[:, 0] >= 0).all() and (p[:, 1] >= 0).all() and (p[:, 2] < 4096).all() and (p[:, 3] < 4096).all()
        assert (p[:, 2] >= p[:, 0]).all() and (p[:, 3] >= p[:, 1]).all()
        assert (p[:, 4] >= 0.1 - 1e-6).all() and (p[:, 4] <= 1.0).all() and np.isin(p[:, 5], [0, 1]).all()
    # (2) stated agreement, on the per-tile detections of the 25-tile batch (small-box filter + class-wise NMS on the GPU,
    # before the ghost-band merge: a random network fires mostly along tile borders, where the merge discards the boxes):
    # a detection that clears the 0.1 score cut by a margin (>= 0.13) in one precision must be found in the other at
    # IoU >= 0.8 with the same class in >= 90 % of the cases; boxes AT the cut may legitimately flip (bf16 moves a score
    # by up to ~2e-2, see (1)), so all boxes are only held to 50 %
    from yolo3 import bbox_utils
    d32 = bbox_utils.detect(r32, min_roi)
    d16 = bbox_utils.detect(r16, min_roi)
    n32 = n16 = 0
    hit = {'c32': [0, 0], 'c16': [0, 0], 'a32': [0, 0], 'a16': [0, 0]}
    for (b32, s32, l32, _), (b16, s16, l16, _) in zip(d32, d16):
        if b32 is None or b16 is None:
            continue
        n32 += len(b32)
        n16 += len(b16)
        for key, (ba, sa, la, bb, lb) in {'32': (b32, s32, l32, b16, l16), '16': (b16, s16, l16, b32, l32)}.items():
            m = _iou_matrix(ba.astype(np.float64), bb.astype(np.float64))
            m = np.where(la[:, None] == lb[None, :], m, 0.0).max(1) >= 0.8
            sure = sa.reshape(-1) >= 0.13
            hit['c' + key][0] += int(m[sure].sum())
            hit['c' + key][1] += int(sure.sum())
            hit['a' + key][0] += int(m.sum())
            hit['a' + key][1] += len(m)
    frac = {k: v[0] / max(v[1], 1) for k, v in hit.items()}
    allsc = np.concatenate([t[1].reshape(-1) for t in d32 if t[0] is not None])
    print('kept scores fp32: min %.3f median %.3f max %.3f' % (allsc.min(), np.median(allsc), allsc.max()))
    print('per-tile detections: fp32 %d, bf16 %d; matched at IoU >= 0.8: confident fp32->bf16 %.3f (%d), bf16->fp32 %.3f (%d); all %.3f / %.3f' % (
        n32, n16, frac['c32'], hit['c32'][1], frac['c16'], hit['c16'][1], frac['a32'], frac['a16']))
    # measured (round 2): 1 098 / 1 136 per-tile detections, median kept score 0.111; confident 50 / 50 matched 1.000 both ways;
    # all boxes 0.595 / 0.575 -- a random network's scores pile up just above the cut, where a bf16 score error of 2e-2 decides
    assert n32 >= 100 and hit['c32'][1] >= 30, 'calibration produced too few detections to compare (%d, %d confident)' % (n32, hit['c32'][1])
    assert frac['c32'] >= 0.9 and frac['c16'] >= 0.9
    assert frac['a32'] >= 0.5 and frac['a16'] >= 0.5
    assert abs(n16 - n32) <= 0.1 * n32


def test_inference_cli_at_416(tmp_path):
    """BASELINE.json configs[0] as far as it exists without TensorFlow / bundled data: inference.py run as a program
    on a folder of 416 x 416 images with a model saved at 416 x 416.  The csv files must equal what the in-process
    pipeline gives for the same images, and the boxes must agree with the CPU oracle's chain (numpy z-score -> oracle
    network fp32 -> decode -> oracle NMS; reference inference.py:40-101) in box IoU."""
    from PIL import Image
    from oracle import model as om
    from oracle import nms as onms
    from yolo3 import bbox_utils, imagereader
    from yolo3.model import YoloV3
    tmp = str(tmp_path)
    img_dir, det_dir = os.path.join(tmp, 'imgs'), os.path.join(tmp, 'dets')
    os.makedirs(img_dir)
    rng = np.random.default_rng(2)
    imgs = []
    for i in range(8):
        yy, xx = np.mgrid[0:416, 0:416]
        im = rng.integers(0, 256, (416, 416, 3)) * 0.5 + 64 + 60 * np.sin(xx / (11.0 + 3 * i))[..., None] * np.cos(yy / 17.0)[..., None]
        im = np.clip(im, 0, 255).astype(np.uint8)
        Image.fromarray(im).save(os.path.join(img_dir, 'a%d.png' % i))
        imgs.append(im)
    x = torch.from_numpy(np.stack([im.astype(np.float32).transpose(2, 0, 1) for im in imgs])).cuda()
    x = imagereader.zscore_normalize_device(x)
    yolo = YoloV3(8, [416, 416, 3], K, ANCHORS, seed=1)
    params = sparse_detector(yolo, x)
    model_file = os.path.join(tmp, 'yolov3.npz')
    yolo.save_weights(model_file)
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + os.environ.get('PYTHONPATH', ''))
    r = subprocess.run([sys.executable, os.path.join(PKG, 'inference.py'), '--saved-model-filepath', model_file, '--output-folder', det_dir,
                        '--image-folder', img_dir, '--image-format', 'png', '--min-box-size', '32'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count('Found:') == 8
    # in-process: the same chain on the same files
    rows = yolo.predict(x)
    dets = bbox_utils.detect(rows, 32, clip_wh=(416, 416))
    net = om.Net(params, 3, len(ANCHORS), K, dtype=torch.float32)
    total, fr_a, fr_b = 0, [], []
    for i, (boxes, scores, lab, _) in enumerate(dets):
        lines = open(os.path.join(det_dir, 'a%d.csv' % i)).read().splitlines()
        assert lines[0] == 'X,Y,W,H,C'
        got = np.array([[int(v) for v in ln.split(',')] for ln in lines[1:]], np.int32).reshape(-1, 5)
        if boxes is None:
            assert got.shape[0] == 0
            continue
        want = np.concatenate((boxes[:, 0:2], boxes[:, 2:4] - boxes[:, 0:2], lab.reshape(-1, 1)), -1).astype(np.int32)
        assert np.array_equal(got, want), 'csv of image %d differs from the in-process pipeline' % i
        total += got.shape[0]
        # the oracle's chain for this image (reference: one image per model call)
        im = imgs[i].astype(np.float32)
        sd = im.std()
        z = (im - im.mean()) / (sd if sd > 1.0 else 1.0)
        with torch.no_grad():
            fms = net.feature_maps(torch.from_numpy(z.transpose(2, 0, 1)[None].astype(np.float32)), training=False)
            orow = om.decode(fms, (416, 416, 3), ANCHORS, K).numpy()[0]
        orow[:, 0:4] = np.clip(orow[:, 0:4], 0, 416)
        keep = onms.detect_rows(orow, 32)
        ob = np.concatenate([orow[k, 0:4] for k in keep])
        oc = np.concatenate([np.full(len(k), c) for c, k in enumerate(keep)])
        fr_a.append(matched_fraction(boxes, lab, ob, oc))
        fr_b.append(matched_fraction(ob, oc, boxes, lab))
    print('416 CLI: %d boxes; matched at IoU >= 0.8 vs the oracle chain: %.3f / %.3f' % (total, np.mean(fr_a), np.mean(fr_b)))
    assert total >= 20, 'calibration produced too few detections (%d)' % total
    assert np.mean(fr_a) >= 0.95 and np.mean(fr_b) >= 0.95
