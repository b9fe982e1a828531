"""GPU tests of the caller-level rows: tiled inference end to end against the reference's
golden, and the three CLIs run as programs on a synthetic lmdb (train -> checkpoint -> export -> inference ->
tiled inference)."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'object-detection-yolov3_amd')


def test_tiled_inference_matches_reference_golden(golden_dir):
    """inference_image_tiled (tiles -> per-tile GPU z-score -> model -> GPU NMS -> ghost-band merge) with the same
    deterministic fake model the reference was run with: the [M,6] prediction array must be IDENTICAL."""
    import inference_tiled
    z = np.load(os.path.join(golden_dir, 'tiled_e2e.npz'))
    rows = z['model_rows']

    class Fake:
        i = 0

        def __call__(self, batch, training=False):
            assert batch.is_cuda and batch.shape[1:] == (1, 608, 608)
            out = rows[self.i:self.i + batch.shape[0]]
            self.i += batch.shape[0]
            return torch.from_numpy(out).cuda()

    img = np.random.default_rng(30).integers(0, 256, tuple(int(v) for v in z['img_shape']), dtype=np.uint8)
    for bs in (1, 5):
        fm = Fake()
        pred = inference_tiled.inference_image_tiled(fm, img, [608, 608], int(z['min_roi']), batch_size=bs)
        assert fm.i == rows.shape[0]
        assert pred.dtype == np.float64 and np.array_equal(pred, z['pred'])


@pytest.mark.parametrize('dtype', [np.uint8, np.uint16, np.float32])
def test_device_tiler_matches_host_tiler(dtype):
    """y3_tile_gather (crop + reflect pad + astype(float32) + HWC -> CHW on the GPU) is bit-identical to
    convert_image_to_tiles, which tests/golden/tiles.json pins to the reference (inference_tiled.py:29-100)."""
    import inference_tiled
    from test_cpu_dataplane import TILE_CASES
    for shape, tile, seed in TILE_CASES + [((1300, 1000, 3), [608, 608], 9)]:
        rng = np.random.default_rng(seed)
        img = (rng.standard_normal(shape) * 50).astype(np.float32) if dtype == np.float32 else rng.integers(0, np.iinfo(dtype).max, shape).astype(dtype)
        tiles, xs, ys = inference_tiled.convert_image_to_tiles(img, tile)
        table, txs, tys = inference_tiled.tile_table(shape[0], shape[1], tile)
        code = inference_tiled._GATHER_DTYPES[np.dtype(dtype)]
        img_dev = torch.from_numpy(img.view(np.int16) if code == 1 else img).cuda()
        table_dev = torch.from_numpy(table).cuda()
        for t0 in range(0, len(tiles), 3):
            cnt = min(3, len(tiles) - t0)
            got = inference_tiled.tiles_to_device(img_dev, code, shape, table_dev, t0, cnt, tile).cpu().numpy()
            want = np.stack([t.astype(np.float32).transpose((2, 0, 1)) for t in tiles[t0:t0 + cnt]])
            assert np.array_equal(got, want), (shape, tile, t0)


@pytest.mark.parametrize('dtype', [np.uint8, np.uint16, np.float32])
def test_fused_tile_preprocessing_equals_the_three_steps(dtype):
    """y3_tile_gather_zscore_nhwc (what inference_tiled feeds the network with) gives the bits of y3_tile_gather ->
    y3_zscore -> NHWC with the channels padded to 4 (the three steps the host path and the model's input transpose perform),
    including the subtract-only branch of the z-score (std <= 1) and a grayscale image."""
    import inference_tiled
    from yolo3 import imagereader
    from yolo3._hip import lib, check
    cases = [((700, 650, 3), [256, 256], 5, 1.0), ((200, 330, 1), [320, 256], 6, 1.0), ((600, 500, 3), [288, 256], 7, 0.004)]
    for shape, tile, seed, gain in cases:
        rng = np.random.default_rng(seed)
        if dtype == np.float32:
            img = (rng.standard_normal(shape) * 50 * gain).astype(np.float32)
        else:
            img = rng.integers(0, max(2, int(np.iinfo(dtype).max * gain)), shape).astype(dtype)
        table, _, _ = inference_tiled.tile_table(shape[0], shape[1], tile)
        code = inference_tiled._GATHER_DTYPES[np.dtype(dtype)]
        img_dev = torch.from_numpy(img.view(np.int16) if code == 1 else img).cuda()
        table_dev = torch.from_numpy(table).cuda()
        n, c = len(table), shape[2]
        assert n > 0
        want = imagereader.zscore_normalize_device(inference_tiled.tiles_to_device(img_dev, code, shape, table_dev, 0, n, tile))
        want = torch.nn.functional.pad(want.permute(0, 2, 3, 1), (0, 4 - c)).contiguous()
        got = torch.full((n, tile[0], tile[1], 4), float('nan'), device='cuda')
        ws = torch.empty(int(lib.y3_zscore_workspace_bytes(n)) // 8 + 1, dtype=torch.float64, device='cuda')
        check(lib.y3_tile_gather_zscore_nhwc(img_dev.data_ptr(), code, shape[0], shape[1], c, table_dev.data_ptr(), n, tile[0], tile[1],
                                             got.data_ptr(), 4, ws.data_ptr(), torch.cuda.current_stream().cuda_stream), 'y3_tile_gather_zscore_nhwc')
        assert torch.equal(got, want), (shape, tile, dtype)


def test_dataset_prefetch_matches_plain_batches(tmp_path):
    """Dataset.batch(n).prefetch(d) (background thread, pinned staging, GPU z-score) yields exactly what the plain batch
    path yields: one in-order reader process, no shuffling; abandoning an iterator stops its producer thread."""
    import threading
    from test_cpu_dataplane import _make_db
    from yolo3.imagereader import ImageReader
    path, _ = _make_db(tmp_path, n=12, size=(64, 64, 3), seed=4)
    anchors = [(64, 384), (384, 64)]
    out = []
    for depth in (0, 2):
        rd = ImageReader(path, anchors, use_augmentation=False, shuffle=False, num_workers=1)
        rd.startup()
        ds = rd.get_tf_dataset().batch(4)
        if depth:
            ds = ds.prefetch(depth)
        it = iter(ds)
        out.append([[t.cpu().clone() for t in next(it)] for _ in range(3)])
        del it
        rd.shutdown()
    for a, b in zip(*out):
        assert len(a) == 4 and all(torch.equal(x, y) for x, y in zip(a, b))
    assert out[0][0][0].shape == (4, 3, 64, 64) and out[0][0][0].is_floating_point()
    deadline = time.time() + 5
    while time.time() < deadline and any(t.name == 'yolo3-prefetch' and t.is_alive() for t in threading.enumerate()):
        time.sleep(0.1)
    assert not any(t.name == 'yolo3-prefetch' and t.is_alive() for t in threading.enumerate())


def test_get_example_is_zscored(tmp_path):
    """The single-example accessors keep the reference's contract (imagereader.py:398,420-436): the image comes out
    z-scored with its own mean / population std (the workers hand out raw pixels for the batched GPU path)."""
    from test_cpu_dataplane import _make_db
    from yolo3.imagereader import ImageReader
    path, truth = _make_db(tmp_path, n=6, size=(64, 64, 3), seed=4)
    rd = ImageReader(path, [(64, 384), (384, 64)], use_augmentation=False, shuffle=False, num_workers=1)
    rd.startup()
    try:
        ex = rd.get_example()
        it = iter(rd.get_tf_dataset())                   # unbatched dataset: same contract
        ex2 = next(it)
    finally:
        rd.shutdown()
    for e, key in ((ex, rd.keys_flat[0]), (ex2, rd.keys_flat[1])):
        raw = truth[key][0].transpose(2, 0, 1).astype(np.float64)
        want = (raw - raw.mean()) / raw.std()
        assert e[0].dtype == np.float32 and abs(float(e[0].mean())) < 1e-5 and abs(float(e[0].std()) - 1.0) < 1e-4
        np.testing.assert_allclose(e[0], want, atol=2e-5)


def _write_dataset(tmp, n, size, K=2, seed=5):
    sys.path.insert(0, PKG)
    import build_lmdb
    from yolo3 import lmdbio
    rng = np.random.default_rng(seed)
    for split, cnt in (('train', n), ('test', max(2, n // 3))):
        items = []
        for i in range(cnt):
            img = rng.integers(0, 256, size, dtype=np.uint8)
            k = int(rng.integers(1, 4))
            wh = rng.integers(40, 120, (k, 2))
            xy = np.stack([rng.integers(0, size[1] - wh[:, 0]), rng.integers(0, size[0] - wh[:, 1])], 1)
            boxes = np.concatenate([xy, wh, rng.integers(0, K, (k, 1))], 1).astype(np.int32)
            items.append(build_lmdb.make_record(img, boxes, i, 'img%03d' % i))
        lmdbio.write_environment(os.path.join(tmp, '%s-syn.lmdb' % split), items)


def test_cli_train_then_inference(tmp_path):
    from PIL import Image
    tmp = str(tmp_path)
    size = (256, 256, 3)
    _write_dataset(tmp, 8, size)
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + os.environ.get('PYTHONPATH', ''))
    out = os.path.join(tmp, 'out')
    r = subprocess.run([sys.executable, os.path.join(PKG, 'train.py'), '--batch_size', '2', '--test_every_n_steps', '3', '--train_database',
                        os.path.join(tmp, 'train-syn.lmdb'), '--test_database', os.path.join(tmp, 'test-syn.lmdb'), '--output_dir', out,
                        '--early_stopping', '1', '--use_augmentation', '1', '--max_epochs', '2', '--learning_rate', '1e-4'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert 'Performing Adam Optimizer learning rate warmup for 3 steps' in r.stdout
    assert r.stdout.count('Train Epoch 0: Batch') == 4            # "step > N: break" runs N+1 steps (Q16)
    losses = [float(v) for v in open(os.path.join(out, 'test_loss.csv')).read().split()]
    assert len(losses) >= 1 and all(np.isfinite(losses))
    model_file = os.path.join(out, 'saved_model', 'yolov3.npz')
    assert os.path.exists(model_file) and os.path.exists(os.path.join(out, 'checkpoint', 'ckpt.npz'))

    # inference.py on a folder of images of the training size
    img_dir, det_dir = os.path.join(tmp, 'imgs'), os.path.join(tmp, 'dets')
    os.makedirs(img_dir)
    rng = np.random.default_rng(1)
    for i in range(2):
        Image.fromarray(rng.integers(0, 256, size, dtype=np.uint8)).save(os.path.join(img_dir, 'a%d.png' % i))
    r = subprocess.run([sys.executable, os.path.join(PKG, 'inference.py'), '--saved-model-filepath', os.path.join(out, 'saved_model'),
                        '--output-folder', det_dir, '--image-folder', img_dir, '--image-format', 'png', '--min-box-size', '8'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    for i in range(2):
        lines = open(os.path.join(det_dir, 'a%d.csv' % i)).read().splitlines()
        assert lines[0] == 'X,Y,W,H,C'
        for ln in lines[1:]:
            x, y, w, h, c = (int(v) for v in ln.split(','))
            assert 0 <= x <= 256 and 0 <= y <= 256 and w > 8 and h > 8 and c in (0, 1)

    # the same folder under a two-rank launcher (replicas only: every rank takes every second image, bf16 conv path)
    det2 = os.path.join(tmp, 'dets2')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', '29533', os.path.join(PKG, 'inference.py'), '--saved-model-filepath', os.path.join(out, 'saved_model'),
                        '--output-folder', det2, '--image-folder', img_dir, '--image-format', 'png', '--min-box-size', '8', '--precision', 'bf16'],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert sorted(os.listdir(det2)) == ['a0.csv', 'a1.csv'] and r.stdout.count('Found:') == 2

    # inference_tiled.py on a larger image, tile = training size
    big_dir, big_out = os.path.join(tmp, 'big'), os.path.join(tmp, 'bigdets')
    os.makedirs(big_dir)
    Image.fromarray(rng.integers(0, 256, (500, 700, 3), dtype=np.uint8)).save(os.path.join(big_dir, 'b.png'))
    r = subprocess.run([sys.executable, os.path.join(PKG, 'inference_tiled.py'), '--saved-model-filepath', os.path.join(out, 'saved_model'),
                        '--output-folder', big_out, '--image-folder', big_dir, '--image-format', 'png', '--tile-height', '256',
                        '--tile-width', '256', '--min-box-size', '8'], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = open(os.path.join(big_out, 'b.csv')).read().splitlines()
    assert lines[0] == 'X,Y,W,H,P,C'
    for ln in lines[1:]:
        x, y, w, h, p, c = ln.split(',')
        assert 0 <= int(x) < 700 and 0 <= int(y) < 500 and 0.1 <= float(p) <= 1.0


def test_weight_file_roundtrip(tmp_path):
    """save_weights / from_file reproduce the model bit for bit (the replacement of the TF checkpoint / SavedModel)."""
    from yolo3.model import YoloV3
    a = YoloV3(2, [64, 64, 3], 2, [(64, 384), (384, 64)], seed=3)
    p = str(tmp_path / 'w.npz')
    a.save_weights(p)
    b = YoloV3.from_file(p)
    x = torch.randn(2, 3, 64, 64, generator=torch.Generator().manual_seed(1)).cuda()
    assert torch.equal(a.predict(x), b.predict(x))
    assert b.anchors == a.anchors and b.img_size == a.img_size and b.number_classes == 2


def test_two_rank_data_parallel_step_on_one_gpu():
    """The N > 1 code path with the real model: two ranks (gloo backend, sharing cuda:0 -- RCCL needs one GPU per rank)
    run bench.py's data-parallel steps with the bucketed async all-reduce hooked into backward; every rank must end
    with bit-identical weights and a finite global loss."""
    import json
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--backend', 'gloo', '--check-replicas', '--bucket-mb', '16'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 16 and out['replicas_identical'] is True
    assert np.isfinite(out['final_loss']) and out['value'] > 0
