"""CPU-only tests: the oracle against the golden vectors produced by the
reference's own code (tests/golden/make_golden.py), the oracle's internal
consistency (fp32 vs fp64, finite differences), and the product's host-side
logic (label layout, CSV writers, tiling) against the same goldens."""
import glob
import json
import os

import numpy as np
import pytest
import torch


def test_oracle_nms_matches_reference_goldens(golden_dir):
    from oracle import nms as onms
    files = [p for p in sorted(glob.glob(os.path.join(golden_dir, 'nms_*.npz'))) if 'units' not in p]
    assert len(files) >= 8
    for path in files:
        z = np.load(path)
        rows, mb = z['rows'], float(z['min_box'])
        f = onms.filter_small_boxes(rows, mb)
        b, s, l = onms.per_class_nms(f[:, 0:4], f[:, 4:5], f[:, 5:])
        if 'keep' not in z:
            assert b is None, path
            assert sum(len(k) for k in onms.detect_rows(rows, mb)) == 0
            continue
        assert np.array_equal(b, z['boxes']) and np.array_equal(s, z['scores']) and np.array_equal(l, z['labels']), path
        keep = np.concatenate(onms.detect_rows(rows, mb))
        assert np.array_equal(keep, z['keep']), path


def test_oracle_nms_units(golden_dir):
    from oracle import nms as onms
    u = np.load(os.path.join(golden_dir, 'nms_units.npz'))
    ious = np.stack([onms.compute_iou(u['boxes'][i], u['boxes']) for i in range(64)])
    assert np.array_equal(np.nan_to_num(ious, nan=-1), np.nan_to_num(u['ious'], nan=-1))
    for thr in (0.3, 0.5, 0.0):
        assert onms.single_class_nms(u['boxes'], u['scores'], thr) == list(u['keep_%g' % thr])
    # IoU exactly at the threshold is kept (iou <= thr): boxes 20 / 21 have IoU 0.5
    assert ious[20, 21] == 0.5


def test_format_boxes_matches_reference(golden_dir):
    """ImageReader.__format_boxes (imagereader.py:252-324) -> product host code."""
    from yolo3.imagereader import format_boxes
    z = np.load(os.path.join(golden_dir, 'labels.npz'))
    for name, K in (('a2k2', 2), ('a3k3', 3)):
        lab = format_boxes(z[name + '_boxes'].copy(), tuple(z[name + '_size']), z[name + '_anchors'], K)
        for i in range(3):
            assert np.array_equal(lab[i], z['%s_label%d' % (name, i + 1)]), (name, i)
        empty = format_boxes(np.zeros((0, 5), np.int32), tuple(z[name + '_size']), z[name + '_anchors'], K)
        assert all(e.sum() == 0 for e in empty)
        assert all(e.sum() == 0 for e in format_boxes(None, tuple(z[name + '_size']), z[name + '_anchors'], K))


def test_csv_writers_match_reference(golden_dir, tmp_path):
    from yolo3 import bbox_utils
    j = json.load(open(os.path.join(golden_dir, 'csv.json')))
    a, b = tmp_path / 'a.csv', tmp_path / 'b.csv'
    bbox_utils.write_boxes_from_xywhc(np.asarray(j['xywhc'], np.int32), str(a))
    bbox_utils.write_boxes_from_ltrbpc(np.asarray(j['ltrbpc'], np.float64), str(b))
    assert a.read_text() == j['xywhc_text']
    assert b.read_text() == j['ltrbpc_text']


def test_layer_table_matches_survey():
    """Architecture walk: 75 convs, 294 trainable tensors, 61,789,770 parameters."""
    from oracle import model as om
    specs = om.layer_specs(3, 2, 2)
    assert len(specs) == 75 and sum(1 for s in specs if s['bn']) == 72
    n = sum(s['k'] ** 2 * s['cin'] * s['cout'] + s['cout'] * (3 if s['bn'] else 1) for s in specs)
    assert n == 61789770
    from yolo3.model import build_layer_specs
    mine, arena, _, _ = build_layer_specs(3, 2, 2)
    assert [(s.cin, s.cout, s.k, s.s, s.bn) for s in mine] == [(s['cin'], s['cout'], s['k'], s['s'], s['bn']) for s in specs]
    assert arena >= n and all(s.w_off % 64 == 0 for s in mine)


def test_same_padding_rule():
    from oracle.model import same_pad
    assert same_pad(416, 3, 1) == (1, 1) and same_pad(416, 3, 2) == (0, 1) and same_pad(13, 3, 2) == (1, 1) and same_pad(7, 1, 1) == (0, 0)


def _tiny(dtype, seed=0, img=64, n=2, training=True):
    from oracle import model as om
    from yolo3.imagereader import format_boxes
    anchors, K = [(64, 384), (384, 64)], 2
    params = om.init_params(3, 2, K, seed=seed)
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(n, 3, img, img, generator=g)
    boxes = np.array([[5, 8, 40, 30, 1], [20, 10, 30, 50, 0]], np.int32)
    labs = [format_boxes(boxes[i:i + 1].copy(), (img, img, 3), anchors, K) for i in range(n)]
    gts = [torch.from_numpy(np.stack([l[s] for l in labs])) for s in range(3)]
    net = om.Net(params, 3, 2, K, dtype=dtype, requires_grad=True)
    return om, net, images.to(dtype), gts, anchors, K, img


def test_oracle_fp32_vs_fp64_forward_and_loss():
    """The restatement evaluated in fp32 and fp64 agrees to fp32 rounding on the forward pass and the loss."""
    out = {}
    for dt in (torch.float32, torch.float64):
        om, net, images, gts, anchors, K, img = _tiny(dt)
        with torch.no_grad():
            fms = net.feature_maps(images, training=True)
            tot = om.compute_loss(fms, gts, (img, img, 3), anchors, K)
            rows = om.decode(fms, (img, img, 3), anchors, K)
        out[dt] = ([f.double() for f in fms], [float(t) for t in tot], rows.double())
    for a, b in zip(out[torch.float32][0], out[torch.float64][0]):
        assert float((a - b).abs().max()) <= 2e-4 * float(b.abs().max())
    np.testing.assert_allclose(out[torch.float32][1], out[torch.float64][1], rtol=1e-4)
    assert out[torch.float64][2].shape == (2, 2 * (4 + 16 + 64), 7)


def test_oracle_loss_gradient_finite_differences():
    """Autograd of the restated loss_layer (model.py:230-354) vs central differences in fp64."""
    from oracle import model as om
    from yolo3.imagereader import format_boxes
    anchors, K, img = [(64, 384), (384, 64)], 2, 64
    g = torch.Generator().manual_seed(3)
    fm = (torch.randn(1, 14, 2, 2, generator=g, dtype=torch.float64) * 0.7).requires_grad_(True)
    gt = torch.from_numpy(format_boxes(np.array([[10, 12, 30, 40, 1]], np.int32), (img, img, 3), anchors, K)[0])[None].double()
    loss = sum(om.loss_layer(fm, gt, (img, img, 3), anchors, K))
    loss.backward()
    eps = 1e-6
    flat = fm.detach().clone().view(-1)
    for idx in range(flat.numel()):
        p, m = flat.clone(), flat.clone()
        p[idx] += eps
        m[idx] -= eps
        lp = sum(om.loss_layer(p.view_as(fm), gt, (img, img, 3), anchors, K))
        lm = sum(om.loss_layer(m.view_as(fm), gt, (img, img, 3), anchors, K))
        fd = float(lp - lm) / (2 * eps)
        assert abs(fd - float(fm.grad.view(-1)[idx])) <= 1e-5 * max(1.0, abs(fd)), idx


def test_oracle_decode_closed_form():
    """A zero feature map decodes to anchor-sized boxes centred in every cell, scores 0.5 (model.py:122-212)."""
    from oracle import model as om
    anchors = [(64, 384), (384, 64)]
    rows = om.decode([torch.zeros(1, 14, g, g) for g in (2, 4, 8)], (64, 64, 3), anchors, 2)
    assert rows.shape == (1, 168, 7)
    r = rows[0, 0]          # scale 1 (stride 32), cell (0,0), anchor 0: centre 16,16, w 64, h 384
    assert torch.allclose(r, torch.tensor([16 - 32.0, 16 - 192.0, 16 + 32.0, 16 + 192.0, 0.5, 0.5, 0.5]))
    r = rows[0, 8 + 2 * (1 * 4 + 2) + 1]   # scale 2 (stride 16), row 1, col 2, anchor 1
    assert torch.allclose(r[:4], torch.tensor([40 - 192.0, 24 - 32.0, 40 + 192.0, 24 + 32.0]))


def test_oracle_adam_is_keras_form():
    """eps sits OUTSIDE the bias correction (App. C5): first step moves every weight by lr*sqrt(1-b2)/(1-b1) * g/(|g|*sqrt(1-b2)+eps)."""
    from oracle import model as om
    p = torch.tensor([1.0, -2.0], dtype=torch.float64)
    st = om.AdamState([p], 0.1)
    g = torch.tensor([1e-3, -5.0], dtype=torch.float64)
    st.step([p], [g])
    lr_t = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    want = np.array([1.0, -2.0]) - lr_t * (0.1 * g.numpy()) / (np.sqrt(0.001 * g.numpy() ** 2) + 1e-7)
    np.testing.assert_allclose(p.numpy(), want, rtol=1e-12)
