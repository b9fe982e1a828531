"""bf16 conv path, network forward only: python tools/bf16_ab.py [--layers]   (A/B builds or switches through the environment)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
sys.path.insert(0, ROOT + '/tools')
import torch         # noqa: E402
import bench         # noqa: E402
from yolo3.model import YoloV3   # noqa: E402
from infer_bench import timed, layer_table   # noqa: E402


def main():
    cases = [(8, 416), (8, 608), (25, 608)]
    if len(sys.argv) > 2 and sys.argv[1] != '--layers':
        cases = [(int(sys.argv[1]), int(sys.argv[2]))]
    for n, img in cases:
        y = YoloV3(n, [img, img, 3], 2, bench.ANCHORS, seed=1, use_graph=True)
        x = torch.randn(n, 3, img, img, generator=torch.Generator().manual_seed(100)).cuda()
        t = timed(lambda: y.predict(x, precision='bf16'), n=20, warm=5)
        fl = bench.conv_flops_per_image(y.specs, img)[0] * n
        print('bf16 forward+decode bs%d %d: %.3f ms  %.1f images/s  %.1f TFLOP/s' % (n, img, t * 1e3, n / t, fl / t / 1e12), flush=True)
        if '--layers' in sys.argv and ((n, img) == (8, 608) or len(cases) == 1):
            y.use_graph = False
            layer_table(y, n, True)
        del y
        torch.cuda.empty_cache()


if __name__ == '__main__':
    main()
