"""End-to-end training throughput with the real data plane: synthetic lmdb (64 uint8 416x416x3 images, 1-4
boxes) -> ImageReader worker processes (lmdb + protobuf decode, augmentation, z-score, label layout) -> batches ->
YoloV3.train_step.  Compare with bench.py (inputs resident).  python tools/train_throughput.py [workers] [augment 0/1]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
import numpy as np   # noqa: E402

workers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
augment = int(sys.argv[2]) if len(sys.argv) > 2 else 1
import build_lmdb                         # noqa: E402
from yolo3 import lmdbio, imagereader     # noqa: E402

tmp = tempfile.mkdtemp()
rng = np.random.default_rng(3)
items = []
for i in range(64):
    img = rng.integers(0, 256, (416, 416, 3), dtype=np.uint8)
    k = int(rng.integers(1, 5))
    wh = rng.integers(40, 300, (k, 2))
    xy = np.stack([rng.integers(0, 416 - np.minimum(wh[:, 0], 415)), rng.integers(0, 416 - np.minimum(wh[:, 1], 415))], 1)
    wh = np.minimum(wh, 416 - xy)
    boxes = np.concatenate([xy, wh, rng.integers(0, 2, (k, 1))], 1).astype(np.int32)
    items.append(build_lmdb.make_record(img, boxes, i, 'img%03d' % i))
path = os.path.join(tmp, 'train-syn.lmdb')
lmdbio.write_environment(path, items)
anchors = [(64, 384), (384, 64)]
reader = imagereader.ImageReader(path, anchors, use_augmentation=bool(augment), shuffle=True, num_workers=workers, balance_classes=True)
reader.startup()                          # worker processes are forked before this process touches the GPU
import torch                              # noqa: E402
from yolo3.model import YoloV3            # noqa: E402
ds = reader.get_tf_dataset().batch(8).prefetch(workers)
yolo = YoloV3(8, reader.get_image_size(), reader.get_number_classes(), anchors, 1e-4)
it = iter(ds)
t_data = t_step = 0.0
n = 0
for step in range(60):
    t0 = time.perf_counter()
    batch = next(it)
    t1 = time.perf_counter()
    loss = yolo.train_step((batch[0], batch[1:4]))
    if step % 10 == 9:
        float(loss)                       # the CLI reads the loss every step; here every 10th to keep the queue full
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if step >= 10:
        t_data += t1 - t0
        t_step += t2 - t1
        n += 1
print('workers %d augment %d: %.1f images/s end to end (waiting for data %.1f ms, step %.1f ms per batch)'
      % (workers, augment, 8 * n / (t_data + t_step), 1e3 * t_data / n, 1e3 * t_step / n))
reader.shutdown()
