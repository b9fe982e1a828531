"""Tiled 4k x 4k inference: end-to-end time per image vs tile batch size (fp32 / bf16).  python tools/tiled_batch.py"""
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
import inference_tiled           # noqa: E402

big = np.random.default_rng(4).integers(0, 256, (4096, 4096, 3), dtype=np.uint8)
for bs in [int(v) for v in (sys.argv[1:] or ["8", "16", "25", "34", "50"])]:
    y = YoloV3(bs, [608, 608, 3], 2, bench.ANCHORS, seed=1, use_graph=True)
    for prec in (('bf16',) if os.environ.get('Y3_SWEEP_BF16_ONLY') else ('fp32', 'bf16')):
        y.inference_precision = prec
        mdl = y.get_keras_model()
        with contextlib.redirect_stdout(io.StringIO()):
            inference_tiled.inference_image_tiled(mdl, big, [608, 608], 32, batch_size=bs)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                inference_tiled.inference_image_tiled(mdl, big, [608, 608], 32, batch_size=bs)
            torch.cuda.synchronize()
            t = (time.perf_counter() - t) / 3
        x = torch.randn(bs, 3, 608, 608).cuda()
        for _ in range(2):
            y.predict(x)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            y.predict(x)
        torch.cuda.synchronize()
        tn = (time.perf_counter() - t1) / 5
        print('batch %2d %s: %.1f ms per 4k image end to end; network %.3f ms per batch = %.1f tiles/s; mem %.1f GB'
              % (bs, prec, t * 1e3, tn * 1e3, bs / tn, torch.cuda.max_memory_allocated() / 2**30), flush=True)
    del y, mdl
    torch.cuda.empty_cache()
