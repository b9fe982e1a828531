"""Soak: N training steps on fresh synthetic batches (host launches, two streams), loss and memory every 100 steps.
python tools/soak.py [steps]"""
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

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 500
yolo = YoloV3(8, [416, 416, 3], 2, bench.ANCHORS, learning_rate=1e-4, seed=1)
g = torch.Generator().manual_seed(7)
pool = [(torch.randn(8, 3, 416, 416, generator=g).cuda(), [torch.from_numpy(x).cuda() for x in bench.synth_labels(np.random.default_rng(s), 8)])
        for s in range(8)]
t0 = time.perf_counter()
for s in range(steps):
    images, gts = pool[s % len(pool)]
    loss = yolo.train_step((images, gts))
    if s % 100 == 99:
        lv = float(loss)
        assert np.isfinite(lv), lv
        print('step %d: loss %.4f, %.1f images/s, allocated %.2f GB, reserved %.2f GB'
              % (s + 1, lv, 8 * (s + 1) / (time.perf_counter() - t0), torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30), flush=True)
out = yolo.predict(pool[0][0], precision='bf16')
assert torch.isfinite(out).all()
print('done: bf16 predict after training finite, %d boxes rows' % out.shape[1])
