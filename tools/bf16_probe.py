"""bf16 forward of 8 x 608^2 (eager launches), a few repetitions -- the command profiled by rocprofv3 for the bf16 conv kernel."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
import torch         # noqa: E402
import bench         # noqa: E402
from yolo3.model import YoloV3   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
y = YoloV3(n, [608, 608, 3], 2, bench.ANCHORS, seed=1)
x = torch.randn(n, 3, 608, 608).cuda()
for _ in range(3):
    y.predict(x, precision='bf16')
torch.cuda.synchronize()
