"""Where the end-to-end time of one tiled 4k x 4k bf16 inference goes (host sections, wall clock):  python tools/tiled_profile.py"""
import contextlib
import io
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
import numpy as np   # noqa: E402
import torch         # noqa: E402
import bench         # noqa: E402
from yolo3.model import YoloV3   # noqa: E402
import inference_tiled as it     # noqa: E402

big = np.random.default_rng(4).integers(0, 256, (4096, 4096, 3), dtype=np.uint8)
y = YoloV3(25, [608, 608, 3], 2, bench.ANCHORS, seed=1, use_graph=True)
y.inference_precision = 'bf16'
mdl = y.get_keras_model()
with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(2):
        it.inference_image_tiled(mdl, big, [608, 608], 32)
torch.cuda.synchronize()
marks = []
orig = {k: getattr(it, k) for k in ('merge_tile_detections', 'finalize_predictions', 'tiles_to_device')}
acc = {k: 0.0 for k in orig}


def wrap(name):
    f = orig[name]

    def g(*a, **k):
        t = time.perf_counter()
        r = f(*a, **k)
        acc[name] += time.perf_counter() - t
        return r
    return g


for k in orig:
    setattr(it, k, wrap(k))
from yolo3 import bbox_utils   # noqa: E402
od = bbox_utils.detect_async
acc['detect_async(queue)'] = 0.0
acc['collect(wait+D2H)'] = 0.0


def da(*a, **k):
    t = time.perf_counter()
    c = od(*a, **k)
    acc['detect_async(queue)'] += time.perf_counter() - t

    def c2():
        t2 = time.perf_counter()
        r = c()
        acc['collect(wait+D2H)'] += time.perf_counter() - t2
        return r
    return c2


bbox_utils.detect_async = da
it.bbox_utils.detect_async = da
with contextlib.redirect_stdout(io.StringIO()):
    t0 = time.perf_counter()
    for _ in range(5):
        it.inference_image_tiled(mdl, big, [608, 608], 32)
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0) / 5
print('end to end %.2f ms per image' % (tot * 1e3))
for k, v in acc.items():
    print('  %-28s %.2f ms per image' % (k, v / 5 * 1e3))
x = torch.randn(45, 3, 608, 608).cuda()
for _ in range(3):
    y.predict(x, precision='bf16')
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5):
    y.predict(x, precision='bf16')
torch.cuda.synchronize()
print('network alone, 45 tiles: %.2f ms -> 100 tiles %.2f ms' % ((time.perf_counter() - t) / 5 * 1e3, (time.perf_counter() - t) / 5 * 1e3 * 100 / 45))
