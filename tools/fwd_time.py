"""fp32 inference forward + decode at batch 8, 416x416 and one training step, timed (development: planner sweeps via env)."""
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

y = YoloV3(8, [416, 416, 3], 2, bench.ANCHORS, seed=1)
x = torch.randn(8, 3, 416, 416, generator=torch.Generator().manual_seed(1)).cuda()
gts = [torch.from_numpy(g).cuda() for g in bench.synth_labels(np.random.default_rng(3), 8)]


def timed(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


ti = timed(lambda: y.predict(x))
tt = timed(lambda: y.train_step((x, gts)), n=15, warm=4)
print('%s | inference %.3f ms (%.0f images/s) | train step %.3f ms (%.1f images/s)' % (
    ' '.join('%s=%s' % (k, v) for k, v in sorted(os.environ.items()) if k.startswith('Y3_')), ti * 1e3, 8 / ti, tt * 1e3, 8 / tt), flush=True)
