"""Probe: inference of a batch as two half-batches on two streams (two model instances = two sets of activation buffers)
vs one launch sequence.  python tools/split_infer_probe.py [batch] [size] [precision]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
import torch         # noqa: E402
import bench         # noqa: E402
from yolo3.model import YoloV3   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
size = int(sys.argv[2]) if len(sys.argv) > 2 else 416
prec = sys.argv[3] if len(sys.argv) > 3 else 'fp32'
a = YoloV3(n, [size, size, 3], 2, bench.ANCHORS, seed=1)
b = YoloV3(n, [size, size, 3], 2, bench.ANCHORS, seed=1)
x = torch.randn(n, 3, size, size).cuda()
h = n // 2
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def whole():
    a.predict(x, precision=prec)


def split():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur)
    s2.wait_stream(cur)
    with torch.cuda.stream(s1):
        a.predict(x[:h], precision=prec)
    with torch.cuda.stream(s2):
        b.predict(x[h:], precision=prec)
    cur.wait_stream(s1)
    cur.wait_stream(s2)


for name, fn in (('whole batch, one stream', whole), ('two halves, two streams', split)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 20
    print('%s: %.3f ms per batch of %d (%s, %d^2) = %.1f images/s' % (name, dt * 1e3, n, prec, size, n / dt), flush=True)
