"""CPU-only tests of the data plane and CLI host logic: protobuf codec vs bytes serialised by the reference's own
generated module, the LMDB reader/writer (round trips: parity with liblmdb is unpinned, none is available), the
ImageReader surface, tiling / tiled-merge host code vs goldens produced by the reference's inference_tiled.py."""
import json
import os
import sys

import numpy as np
import pytest


def test_protobuf_codec_matches_reference_bytes(golden_dir):
    from yolo3.isg_ai_pb import ImageYoloBoxesPair
    z = np.load(os.path.join(golden_dir, 'proto_pair.npz'))
    assert ImageYoloBoxesPair.from_arrays(z['img'], z['boxes']).SerializeToString() == z['msg'].tobytes()
    assert ImageYoloBoxesPair.from_arrays(z['img1'], None).SerializeToString() == z['msg1'].tobytes()
    m = ImageYoloBoxesPair().ParseFromString(z['msg'].tobytes())
    img, boxes = m.to_arrays()
    assert np.array_equal(img, z['img']) and np.array_equal(boxes, z['boxes'])
    assert (m.channels, m.img_height, m.img_width, m.box_count, m.img_type, m.box_type) == (3, 6, 5, 2, '|u1', '<i4')
    m1 = ImageYoloBoxesPair().ParseFromString(z['msg1'].tobytes())
    img1, boxes1 = m1.to_arrays()
    assert np.array_equal(img1, z['img1']) and boxes1.shape == (0, 5)


@pytest.mark.parametrize('n,vsize', [(0, 0), (1, 10), (50, 100), (3000, 40), (40, 20000), (700, 5000)])
def test_lmdb_round_trip(tmp_path, n, vsize):
    """Leaf-only, multi-level branch trees and overflow values; ordered iteration and point lookups."""
    from yolo3 import lmdbio
    rng = np.random.default_rng(n + vsize)
    items = {('%d_img%04d:%s' % (i, i, '0,1' if i % 3 else '')).encode(): rng.integers(0, 256, vsize + (i % 7), dtype=np.uint8).tobytes() for i in range(n)}
    path = str(tmp_path / 'db.lmdb')
    assert lmdbio.write_environment(path, items.items()) == n
    assert os.path.exists(os.path.join(path, 'data.mdb')) and os.path.exists(os.path.join(path, 'lock.mdb'))
    with lmdbio.Environment(path) as env:
        assert env.stat()['entries'] == n
        keys = list(env.keys())
        assert keys == sorted(items)                  # LMDB's bytewise order: b'10_...' < b'2_...'
        got = dict(env.items())
        assert got == items
        for k in list(items)[::max(1, n // 17)]:
            assert env.get(k) == items[k]
        assert env.get(b'missing') is None
        if n >= 700:
            assert env.stat()['depth'] >= 2


def test_lmdb_rejects_garbage(tmp_path):
    from yolo3 import lmdbio
    p = tmp_path / 'x.lmdb'
    p.mkdir()
    (p / 'data.mdb').write_bytes(b'\0' * 8192)
    with pytest.raises(lmdbio.LmdbError):
        lmdbio.Environment(str(p))
    with pytest.raises(lmdbio.LmdbError):
        lmdbio.Environment(str(tmp_path / 'nope'))


def _make_db(tmp_path, n=12, size=(64, 64, 3), K=2, with_empty=True, seed=3):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'object-detection-yolov3_amd'))
    import build_lmdb
    from yolo3 import lmdbio
    rng = np.random.default_rng(seed)
    items, truth = [], {}
    for i in range(n):
        img = rng.integers(0, 256, size, dtype=np.uint8)
        k = 0 if (with_empty and i % 5 == 4) else int(rng.integers(1, 4))
        wh = rng.integers(12, 40, (k, 2))
        xy = np.stack([rng.integers(0, size[1] - wh[:, 0]), rng.integers(0, size[0] - wh[:, 1])], 1) if k else np.zeros((0, 2), int)
        boxes = np.concatenate([xy, wh, rng.integers(0, K, (k, 1))], 1).astype(np.int32)
        key, val = build_lmdb.make_record(img, boxes, i, 'img%03d' % i)
        items.append((key, val))
        truth[key] = (img, boxes)
    path = str(tmp_path / 'train-x.lmdb')
    lmdbio.write_environment(path, items)
    return path, truth


def test_image_reader_surface(tmp_path):
    """ImageReader (imagereader.py:79-460): key parsing, class bookkeeping, example formatting, worker processes."""
    from yolo3.imagereader import ImageReader, format_boxes
    anchors = [(64, 384), (384, 64)]
    path, truth = _make_db(tmp_path)
    rd = ImageReader(path, anchors, use_augmentation=False, shuffle=False, num_workers=2)
    assert rd.get_image_size() == [64, 64, 3] and rd.get_image_count() == 12
    assert rd.get_number_classes() == 2               # classes 0/1 plus the bucket of box-free images
    assert len(rd.keys) == 3 and sum(len(k) for k in rd.keys) >= 12
    assert rd.keys_flat == sorted(truth)              # lmdb key order
    from yolo3 import lmdbio
    with lmdbio.Environment(path) as env:
        key = rd.keys_flat[3]
        img, l1, l2, l3 = rd.load_example(key, env)
    timg, tboxes = truth[key]
    assert img.dtype == np.float32 and img.shape == (3, 64, 64) and np.array_equal(img, timg.transpose(2, 0, 1).astype(np.float32))
    want = format_boxes(tboxes.copy(), (64, 64, 3), anchors, 2)
    assert all(np.array_equal(a, b) for a, b in zip((l1, l2, l3), want))
    assert l1.shape == (2, 2, 2, 7) and l3.shape == (8, 8, 2, 7)
    # worker processes + bounded queue + orderly shutdown
    rd.startup()
    gen = rd.raw_generator()                          # generator() / get_example() z-score on the GPU: tests/test_gpu_cli.py
    seen = [next(gen) for _ in range(7)]
    assert all(s[0].shape == (3, 64, 64) for s in seen)
    rd.shutdown()
    with pytest.raises(Exception):
        ImageReader(str(tmp_path / 'absent.lmdb'), anchors)


def test_unshuffled_reader_shards_are_disjoint(tmp_path):
    """One reader per rank (train.py under torch.distributed.run): worker w of shard s starts at s * workers + w and strides by
    shards * workers, so the ranks' test readers together walk the key list once, without overlap."""
    from yolo3.imagereader import ImageReader
    path, _ = _make_db(tmp_path, n=12, size=(64, 64, 3), seed=2)
    anchors = [(64, 384), (384, 64)]
    walked = []
    for shard in range(2):
        rd = ImageReader(path, anchors, use_augmentation=False, shuffle=False, num_workers=3, num_shards=2, shard_index=shard)
        for w in range(3):
            state = {'idx': rd.shard_index * rd.nb_workers + w}
            walked += [rd._next_key(state) for _ in range(2)]
        ds = rd.get_tf_dataset().shard(2, shard)          # consistent request: accepted
        assert ds.reader is rd
    assert sorted(walked) == rd.keys_flat and len(set(walked)) == 12
    rd.set_shard(2, 0)                                    # not started yet: may still change
    assert rd.shard_index == 0


def test_image_reader_augmentation_keeps_contract(tmp_path):
    from yolo3.imagereader import ImageReader
    from yolo3 import lmdbio
    path, truth = _make_db(tmp_path, n=6, size=(96, 96, 1), with_empty=False, seed=9)
    rd = ImageReader(path, [(64, 384), (384, 64)], use_augmentation=True, shuffle=True, balance_classes=True, num_workers=1)
    np.random.seed(0)
    with lmdbio.Environment(path) as env:
        for key in rd.keys_flat * 3:
            img, l1, l2, l3 = rd.load_example(key, env)
            assert img.shape == (1, 96, 96) and img.dtype == np.float32 and np.isfinite(img).all()
            assert l1.shape == (3, 3, 2, 7) and l3.shape == (12, 12, 2, 7)
            on = l3[..., 4] > 0
            assert (l3[on][:, 2:4] > 0).all()


def test_convert_image_to_tiles_matches_reference(golden_dir):
    import inference_tiled
    j = json.load(open(os.path.join(golden_dir, 'tiles.json')))
    for name, g in j.items():
        h, w, c = g['shape']
        img = np.random.default_rng(g['seed']).integers(0, 256, (h, w, c), dtype=np.uint8)
        tiles, xs, ys = inference_tiled.convert_image_to_tiles(img, g['tile'])
        assert xs == g['x'] and ys == g['y'], name
        assert [list(t.shape) for t in tiles] == g['tile_shapes'], name
        assert [int(t.astype(np.int64).sum()) for t in tiles] == g['sums'], name
        assert [t[0:2, 0:2, 0].astype(int).tolist() for t in tiles] == g['corner'], name


def _tile_from_table(img, row, tile):
    """Host statement of y3_tile_gather's indexing (reflect_index in pointwise.hip): crop-relative periodic reflection."""
    def idx(n_out, start, n, pre):
        p = np.arange(n_out) - pre
        if n == 1:
            return np.full(n_out, start)
        period = 2 * (n - 1)
        q = np.mod(p, period)
        return start + np.where(q < n, q, period - q)
    y0, ny, pre_y, x0, nx, pre_x = [int(v) for v in row]
    return img[idx(tile[0], y0, ny, pre_y)][:, idx(tile[1], x0, nx, pre_x)]


TILE_CASES = [((417, 1250, 3), [608, 608], 1), ((833, 420, 1), [608, 512], 2), ((300, 200, 3), [512, 512], 3), ((513, 609, 2), [512, 608], 4),
              ((1249, 418, 1), [608, 608], 5)]


def test_tile_table_reproduces_reference_tiles(golden_dir):
    """tile_table + the device tiler's reflect rule == convert_image_to_tiles (which the goldens pin to the reference),
    including crops shorter than their padding (np.pad reflects repeatedly) and images smaller than one tile."""
    import inference_tiled
    j = json.load(open(os.path.join(golden_dir, 'tiles.json')))
    cases = [(g['shape'], g['tile'], g['seed']) for g in j.values()] + TILE_CASES
    for shape, tile, seed in cases:
        img = np.random.default_rng(seed).integers(0, 256, tuple(shape), dtype=np.uint8)
        tiles, xs, ys = inference_tiled.convert_image_to_tiles(img, tile)
        table, txs, tys = inference_tiled.tile_table(shape[0], shape[1], tile)
        assert txs == xs and tys == ys and len(table) == len(tiles)
        assert all(t.shape[:2] == tuple(tile) for t in tiles), [t.shape for t in tiles]
        for t, row in zip(tiles, table):
            assert np.array_equal(_tile_from_table(img, row, tile), t), (shape, tile, row)


def test_tiled_merge_matches_reference(golden_dir):
    """Ghost-band rejection, global shift, rounding and clamping (inference_tiled.py:230-310) against the reference
    run end to end with a deterministic fake model; the per-tile NMS comes from the oracle here (GPU twin: test_gpu_cli.py)."""
    import inference_tiled
    from oracle import nms as onms
    z = np.load(os.path.join(golden_dir, 'tiled_e2e.npz'))
    img_shape, tile = tuple(int(v) for v in z['img_shape']), [int(v) for v in z['tile']]
    tiles, xs, ys = inference_tiled.convert_image_to_tiles(np.zeros(img_shape, np.uint8), tile)
    assert len(tiles) == z['model_rows'].shape[0]
    bl, sl, cl = [], [], []
    for i, rows in enumerate(z['model_rows']):
        f = onms.filter_small_boxes(rows, int(z['min_roi']))
        b, s, l = onms.per_class_nms(f[:, 0:4], f[:, 4:5], f[:, 5:])
        if b is None:
            continue
        r = inference_tiled.merge_tile_detections(b, s, l, xs[i], ys[i], tile, img_shape)
        if r is not None:
            bl.append(r[0]); sl.append(r[1]); cl.append(r[2])
    pred = inference_tiled.finalize_predictions(bl, sl, cl, img_shape)
    assert pred.shape == z['pred'].shape and np.array_equal(pred, z['pred'])


def test_load_boxes_csv(tmp_path):
    from yolo3 import bbox_utils
    p = tmp_path / 'a.csv'
    p.write_text('X, Y, W, H, C\n1, 2, 3, 4, 0\n10,20,30,40,1\n')
    a = bbox_utils.load_boxes_to_xywhc(str(p))
    assert a.tolist() == [[1, 2, 3, 4, 0], [10, 20, 30, 40, 1]]
    assert bbox_utils.load_boxes_to_ltrbc(str(p)).tolist() == [[1, 2, 3, 5, 0], [10, 20, 39, 59, 1]]
    assert bbox_utils.load_boxes_to_xywhc(str(tmp_path / 'none.csv')).shape == (0, 5)


def test_find_anchor_sizes_recovers_clusters(tmp_path, capsys):
    """find_anchor_sizes.py:19-51: (H, W) of every box of every csv, k = 2..7, prints score (= -inertia) and centres."""
    import find_anchor_sizes
    from yolo3 import bbox_utils
    rng = np.random.default_rng(0)
    true = np.array([[40, 30], [120, 200], [300, 280]])             # (H, W)
    for f in range(4):
        rows = []
        for c, (h, w) in enumerate(true):
            for _ in range(25):
                rows.append([int(rng.integers(0, 500)), int(rng.integers(0, 500)), int(w + rng.integers(-5, 6)), int(h + rng.integers(-5, 6)), c])
        bbox_utils.write_boxes_from_xywhc(np.asarray(rows, np.int32), str(tmp_path / ('a%d.csv' % f)))
    X = find_anchor_sizes.load_sizes(str(tmp_path))
    assert X.shape == (300, 2)
    out = find_anchor_sizes.find_anchors(str(tmp_path), seed=1, plot=False)
    c3 = out[3][np.argsort(out[3][:, 0])]
    assert np.abs(c3 - true).max() < 2.0
    text = capsys.readouterr().out
    assert text.count('score for') == 6 and 'score for 3-means = -' in text
    centers, labels, inertia = find_anchor_sizes.kmeans(X, 3, np.random.default_rng(2))
    assert abs(inertia - sum(((X[labels == j] - centers[j]) ** 2).sum() for j in range(3))) < 1e-6
    with pytest.raises(ValueError):
        find_anchor_sizes.kmeans(X[:2], 3, rng)


def test_augment_matches_reference_goldens(golden_dir):
    """yolo3/augment.py against tests/golden/augment.npz, produced by running the reference's augment.py under seeded
    np.random (make_golden.py:g8_augment): box jitter, box affine helper and crop offsets IDENTICAL; un-rescaled pixels
    (incl. noise and blur) identical; rescaled pixels within 5e-3 on the 0..255 range (skimage's rescale restated on
    scipy.ndimage.map_coordinates)."""
    from yolo3 import augment
    z = np.load(os.path.join(golden_dir, 'augment.npz'))
    boxes = z['boxes']
    np.random.seed(123)
    assert np.array_equal(augment.jitter_boxes(boxes.copy(), 0.05, 0.08, (260, 420)), z['jitter'])
    for i, (c, crop) in enumerate(zip(z['affine_cases'], z['affine_crops'])):
        rx, ry, sx, sy, dx, dy = c
        r = augment.transform_boxes(boxes.copy(), tuple(crop), bool(rx), bool(ry), sx, sy, int(dx), int(dy))
        want = z['affine_boxes_%d' % i]
        assert (r is None and want.shape[0] == 0) or np.array_equal(r, want), i
    for i, (c, crop) in enumerate(zip(z['affine_img_cases'], z['affine_img_crops'])):
        rx, ry, sx, sy = c
        np.random.seed(50 + i)
        img, dx, dy = augment.transform_image(z['img'], bool(rx), bool(ry), sx, sy, tuple(crop))
        assert [dx, dy] == z['affine_img_%d_dxdy' % i].tolist()
        assert np.abs(img - z['affine_img_%d' % i]).max() <= (1e-6 if sx == 1 and sy == 1 else 5e-3), i
    np.random.seed(60)
    img, dx, dy = augment.transform_image(z['img2'], True, False, 1.2, 0.95, (60, 90))          # 2-D image
    assert [dx, dy] == z['affine_img2_dxdy'].tolist() and np.abs(img - z['affine_img2']).max() <= 5e-3
    base = dict(reflection_flag=True, crop_to=(64, 96), noise_augmentation_severity=0.02, blur_augmentation_max_sigma=2,
                box_size_augmentation_severity=0.03, box_location_jitter_severity=0.03)
    for i, scale in enumerate((0, 0.1)):
        for seed in (1, 2, 3):
            np.random.seed(seed)
            img, b = augment.augment_image_box_pair(z['img'].copy(), z['pair_boxes'].copy(), scale_augmentation_severity=scale, **base)
            want = z['pair_%d_%d_boxes' % (i, seed)]
            assert (b is None and want.shape[0] == 0) or np.array_equal(b, want), (i, seed)
            assert img.dtype == np.float32 and np.abs(img - z['pair_%d_%d_img' % (i, seed)]).max() <= (0.0 if scale == 0 else 5e-3), (i, seed)
    with pytest.raises(AssertionError):                                  # augment.py:180: a box jittered to nothing aborts
        np.random.seed(0)
        augment.jitter_boxes(np.asarray([[500, 10, 3, 3, 0]], np.int32), 0.0, 0.0, (100, 100))


def test_ltrbc_writer_and_inverse_format_boxes_match_reference(golden_dir, tmp_path):
    """bbox_utils.write_boxes_from_ltrbc text and imagereader.inverse_format_boxes (anchor-0 boxes of a label tensor)
    against the reference's outputs (make_golden.py:g9_misc)."""
    from yolo3 import imagereader
    try:
        from yolo3 import bbox_utils
    except Exception as e:                       # the module binds libyolo3hip.so at import; built by __graft_entry__.build()
        pytest.skip('libyolo3hip.so not built: %s' % e)
    j = json.load(open(os.path.join(golden_dir, 'misc.json')))
    fp = str(tmp_path / 'a.csv')
    bbox_utils.write_boxes_from_ltrbc(np.asarray(j['ltrbc'], np.int32), fp)
    assert open(fp).read() == j['ltrbc_text']
    l1, l2, l3 = imagereader.format_boxes(np.asarray(j['fmt_boxes'], np.int32), [416, 416, 3], [(64, 384), (384, 64)], 2)
    assert np.argwhere(l3[..., 4] > 0).tolist() == j['label3_nonzero']
    inv = imagereader.inverse_format_boxes(np.stack([l3, l3]), 1)
    assert np.asarray(inv).tolist() == j['inverse']


def test_upload_bands_cover_every_batch():
    """inference_tiled.upload_bands: every tile of batch i lies inside the rows uploaded up to batch i, the bands never shrink, the
    last one is the whole image (the network of batch 0 starts while the rest of the image is still on its way)."""
    import inference_tiled as it
    for (h, w), tile in (((4096, 4096), [608, 608]), ((1300, 1000), [608, 608]), ((700, 5000), [320, 416]), ((500, 400), [608, 608])):
        table, _, _ = it.tile_table(h, w, tile)
        n = len(table)
        for sizes in ([n], it.plan_tile_batches(n, tile), [1] * n, [max(1, n // 3)] * 3 + [n - 3 * max(1, n // 3)] if n >= 3 else [n]):
            sizes = [s for s in sizes if s > 0]
            if sum(sizes) != n:
                continue
            bands = it.upload_bands(table, sizes, h)
            assert len(bands) == len(sizes) and bands[-1] == h and all(a <= b for a, b in zip(bands, bands[1:]))
            t0 = 0
            for nb, hi in zip(sizes, bands):
                assert int((table[t0:t0 + nb, 0] + table[t0:t0 + nb, 1]).max()) <= hi <= h
                t0 += nb
    table, _, _ = it.tile_table(4096, 4096, [608, 608])
    assert it.upload_bands(table, [45, 45, 10], 4096)[0] < 4096 * 0.55      # the first 45 of 100 tiles need about half of the rows


def test_tile_batch_planner():
    """inference_tiled.plan_tile_batches (host logic of the bf16 tiled path): batches cover every tile once, whole rounds of
    256 workgroups at the 256 x 256-tile stages, small images stay one batch."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('inference_tiled_for_test', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'object-detection-yolov3_amd', 'inference_tiled.py'))
    it = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(it)
    except Exception as e:                   # the module imports the HIP binding at the top: nothing to test without the library
        pytest.skip('inference_tiled not importable here: %s' % e)
    assert it.plan_tile_batches(100, [608, 608]) == [45, 45, 10]
    for n in (0, 1, 7, 9, 16, 36, 64, 99, 100, 101, 400):
        for tile in ([608, 608], [512, 512], [416, 416], [608, 352]):
            b = it.plan_tile_batches(n, tile)
            assert sum(b) == n and all(0 < v <= 64 for v in b) and len(set(b[:-1])) <= 1
    # 45 tiles of 608^2: 1016 / 508 / 256 workgroups = 3.97 / 1.98 / 1.0 rounds of 256
    assert [-(-(45 * (608 // s) ** 2) // 256) * c for s, c in ((8, 1), (16, 2), (32, 4))] == [1016, 508, 256]
