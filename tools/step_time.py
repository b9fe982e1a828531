"""Plain step timing (no instrumentation): python tools/step_time.py [graph 0/1]"""
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

graph = int(sys.argv[1]) if len(sys.argv) > 1 else 1
yolo = YoloV3(8, [416, 416, 3], 2, bench.ANCHORS, learning_rate=1e-4, seed=1, use_graph=bool(graph))
images = torch.randn(8, 3, 416, 416, generator=torch.Generator().manual_seed(100)).cuda()
gts = [torch.from_numpy(x).cuda() for x in bench.synth_labels(np.random.default_rng(3), 8)]
for _ in range(5):
    loss = yolo.train_step((images, gts))
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(20):
    loss = yolo.train_step((images, gts))
t_enq = (time.perf_counter() - t) / 20          # host time to enqueue a step (the queue may push back when it is full)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 20
print('graph %d: %.3f ms per step, %.1f images/s, loss %.6f; host enqueue %.3f ms per step' % (graph, dt * 1e3, 8 / dt, float(loss), t_enq * 1e3))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    yolo.train_step((images, gts))
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(12)
