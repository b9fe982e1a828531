"""Where the HOST spends its time while it queues training steps (cProfile over 10 steps, no synchronisation inside).
Run on the GPU box:  python tools/host_profile.py"""
import cProfile
import os
import pstats
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
for _ in range(4):
    y.train_step((x, gts))
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    y.train_step((x, gts))
ti = time.perf_counter() - t
torch.cuda.synchronize()
print('host issue %.2f ms per step, with the GPU %.2f ms per step' % (ti * 100, (time.perf_counter() - t) * 100))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    y.train_step((x, gts))
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('tottime').print_stats(18)

# one step at a time, queue drained before each: the host cost of a step without back-pressure from a full queue
ts = []
for _ in range(5):
    torch.cuda.synchronize()
    t = time.perf_counter()
    y.train_step((x, gts))
    ts.append((time.perf_counter() - t) * 1e3)
    torch.cuda.synchronize()
print('host issue, one step with an empty queue: %s ms' % ['%.2f' % v for v in ts])
