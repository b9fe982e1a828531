"""Driver for the rocprofv3 passes over the HBM-bound part of the path (north_star: "achieved HBM GB/s for the decode/NMS
path"): z-score -> network -> decode -> class-wise NMS at inference bs 8 x 416^2 (BASELINE config 1) and 25 x 608^2
(one launch of the tiled path, config 4), NMS on the SURVEY 8d stress rows (sparse and dense), and one training step
(Adam).  Run it directly after `--`:

    rocprofv3 --kernel-trace --stats -d OUT/trace -o t --output-format csv -- python3 tools/hbm_path_driver.py
    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d OUT/fetch -o f --output-format csv -- python3 tools/hbm_path_driver.py
    rocprofv3 --kernel-trace --pmc WRITE_SIZE -d OUT/write -o w --output-format csv -- python3 tools/hbm_path_driver.py
    python tools/hbm_path_summary.py OUT/trace/..kernel_trace.csv OUT/fetch/..counter_collection.csv OUT/write/..counter_collection.csv

Every shape is run REPS times after a warm-up; the summary averages per (kernel, grid size)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, ROOT + '/object-detection-yolov3_amd')
import numpy as np   # noqa: E402
import torch         # noqa: E402
import bench         # noqa: E402
from yolo3.model import YoloV3          # noqa: E402
from yolo3 import bbox_utils, imagereader   # noqa: E402

REPS = 10


def stress_rows(seed, n, nb, K, img, dense):
    """SURVEY 8d 'NMS stress' rows [n, nb, 5+K] float32."""
    rng = np.random.default_rng(seed)
    cx, cy = rng.uniform(0, img, (n, nb)), rng.uniform(0, img, (n, nb))
    w, h = rng.uniform(33, 300, (n, nb)), rng.uniform(33, 300, (n, nb))
    obj = rng.uniform(0, 1, (n, nb))
    if not dense:
        obj = obj ** 8
    cls = rng.uniform(0, 1, (n, nb, K))
    rows = np.concatenate([np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, obj], -1), cls], -1).astype(np.float32)
    return torch.from_numpy(rows).cuda()


def main():
    torch.cuda.set_device(0)
    K, anchors = bench.K, bench.ANCHORS
    for n, img in ((8, 416), (25, 608)):
        y = YoloV3(n, [img, img, 3], K, anchors, seed=1)
        g = torch.Generator().manual_seed(7)
        raw = torch.randint(0, 256, (n, 3, img, img), generator=g).float().cuda()      # what the reader hands over: pixel values
        for _ in range(2 + REPS):
            x = imagereader.zscore_normalize_device(raw)
            rows = y.predict(x)
            bbox_utils.nms_device(rows, 32.0, clip_wh=(img, img))
        torch.cuda.synchronize()
        nb = rows.shape[1]
        for dense in (False, True):
            r = stress_rows(0, n, nb, K, img, dense)
            for _ in range(2 + REPS):
                bbox_utils.nms_device(r, 32.0, clip_wh=(img, img))
            torch.cuda.synchronize()
        if img == 416:
            images = torch.randn(n, 3, img, img, generator=g).cuda()
            gts = [torch.from_numpy(v).cuda() for v in bench.synth_labels(np.random.default_rng(3), n)]
            for _ in range(2 + REPS):
                y.train_step((images, gts))
            torch.cuda.synchronize()
        del y
        torch.cuda.empty_cache()
    print('hbm_path_driver done')


if __name__ == '__main__':
    main()
