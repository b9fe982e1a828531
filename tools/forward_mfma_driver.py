"""Driver for the whole-forward MFMA utilisation (north_star: >= 40 % on the Darknet-53 forward at batch 8, 416 x 416):
N fp32 inference forwards (forward + decode, host launches) under rocprofv3 --pmc; tools/forward_mfma_util.py sums the counters.
python tools/forward_mfma_driver.py [forwards]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
import torch      # noqa: E402
import bench      # noqa: E402
from yolo3.model import YoloV3   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
y = YoloV3(8, [416, 416, 3], 2, bench.ANCHORS, seed=1, use_graph=False)
x = torch.randn(8, 3, 416, 416).cuda()
for _ in range(n):
    y.predict(x)
torch.cuda.synchronize()
print('forwards', n)
