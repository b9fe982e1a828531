"""Inference throughput, fp32 vs bf16 conv path: bs=8 416x416 (BASELINE config 1 shape) and the tiled 4k x 4k case
(config 4: 608x608 tiles).  Run on the GPU box:  python tools/infer_bench.py [--layers]"""
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
from yolo3._hip import check     # noqa: E402
import inference_tiled           # noqa: E402


def timed(fn, n=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def layer_table(yolo, n, bf16):
    plan = yolo._plan(n, False, bf16)
    st = torch.cuda.current_stream().cuda_stream
    best = {}
    for rep in range(3):
        recs = []
        for fn, args in plan.fwd:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            check(fn(*args, st), 'x')
            b.record()
            recs.append((fn.__name__, args, a, b))
        torch.cuda.synchronize()
        for i, (nm, args, a, b) in enumerate(recs):
            best[i] = min(best.get(i, 1e30), a.elapsed_time(b) * 1e3)
    agg, tot = {}, {}
    for i, (nm, args, a, b) in enumerate(recs):
        tot[nm] = tot.get(nm, 0) + best[i]
        if nm in ('y3_conv2d_fwd', 'y3_conv2d_fwd_bf16', 'y3_conv2d_fwd_bf16_ws'):
            src, dst, k, s = args[0], args[5], args[3], args[4]
            key = (dst.n * dst.h * dst.w, src.c, dst.c, k, s)
            d = agg.setdefault(key, [0, 0.0])
            d[0] += 1
            d[1] += best[i]
    print('%8s %5s %5s k s  cnt   total_us   avg_us  TFLOP/s' % ('M', 'cin', 'cout'))
    for key, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        m, cin, cout, k, s = key
        print('%8d %5d %5d %d %d %4d %10.1f %8.1f %8.1f' % (m, cin, cout, k, s, d[0], d[1], d[1] / d[0], 2.0 * m * k * k * cin * cout * d[0] / d[1] / 1e6))
    print('totals (us):', {k: round(v, 1) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])}, 'sum', round(sum(tot.values()), 1))


def main():
    yolo = YoloV3(8, [416, 416, 3], 2, bench.ANCHORS, seed=1)
    images = torch.randn(8, 3, 416, 416, generator=torch.Generator().manual_seed(100)).cuda()
    fl = bench.conv_flops(yolo.specs, 416, 416)[0] if hasattr(bench, 'conv_flops') else None
    for prec in ('fp32', 'bf16'):
        t = timed(lambda: yolo.predict(images, precision=prec))
        print('bs8 416 %s: %.3f ms  %.1f images/s' % (prec, t * 1e3, 8 / t), flush=True)
    if '--layers' in sys.argv:
        layer_table(yolo, 8, True)
    del yolo
    torch.cuda.empty_cache()
    y608 = YoloV3(8, [608, 608, 3], 2, bench.ANCHORS, seed=1)
    big = np.random.default_rng(5).integers(0, 255, (4096, 4096, 3), dtype=np.uint8)
    for prec in ('fp32', 'bf16'):
        y608.inference_precision = prec
        mdl = y608.get_keras_model()
        t = timed(lambda: inference_tiled.inference_image_tiled(mdl, big, [608, 608], 32), n=3, warm=1)
        x = torch.randn(8, 3, 608, 608).cuda()
        tn = timed(lambda: y608.predict(x))
        print('tiled 4096x4096, 608 tiles, %s: %.1f ms per image end to end (100 tiles); network only bs8 608: %.3f ms = %.1f tiles/s'
              % (prec, t * 1e3, tn * 1e3, 8 / tn), flush=True)
    if '--layers' in sys.argv:
        layer_table(y608, 8, True)


if __name__ == '__main__':
    main()
