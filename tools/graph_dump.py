"""Development tool: capture one training step as a HIP graph with the loss entry's flag clear as
hipMemsetAsync (the round-3 form: DEV library, Y3_LOSS_MEMSET=1) and with the shipped clear kernel, dump both graphs
(hipGraphDebugDotPrint through torch.cuda.CUDAGraph.debug_dump) and report what every memset / memcpy node is wired to.
    Y3_LIB=object-detection-yolov3_amd/yolo3/_lib/libyolo3hip_dev.so Y3_LOSS_MEMSET=1 python tools/graph_dump.py out_prefix [img] [batch]"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/object-detection-yolov3_amd']
import numpy as np   # noqa: E402
import torch         # noqa: E402
import bench         # noqa: E402
from yolo3.model import YoloV3   # noqa: E402

prefix = sys.argv[1]
img = int(sys.argv[2]) if len(sys.argv) > 2 else 416
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2
bench.IMG = img
y = YoloV3(n, [img, img, 3], 2, bench.ANCHORS, learning_rate=1e-4, seed=1, use_graph=True)
images = torch.randn(n, 3, img, img, generator=torch.Generator().manual_seed(1)).cuda()
gts = [torch.from_numpy(x).cuda() for x in bench.synth_labels(np.random.default_rng(3), n)]
plan = y._plan(n, True)
y._load_inputs(plan, images, gts)
y.lr_t_dev.fill_(1e-4)
st = y._stream()
plan.run_forward(st)
plan.run_loss(st)
plan.run_backward(st)
torch.cuda.synchronize()
# capture by hand (torch.cuda.CUDAGraph.debug_dump wrote no file on this ROCm build): a stream of our own, made torch's current
# stream too, hipStreamBeginCapture / EndCapture around exactly what YoloV3._capture records, then hipGraphDebugDotPrint
import ctypes
hip = ctypes.CDLL('libamdhip64.so')
stream = ctypes.c_void_p()
assert hip.hipStreamCreate(ctypes.byref(stream)) == 0
ext = torch.cuda.ExternalStream(stream.value)
graph = ctypes.c_void_p()
with torch.cuda.stream(ext):
    assert hip.hipStreamBeginCapture(stream, 2) == 0          # hipStreamCaptureModeRelaxed
    st = y._stream()
    assert st == stream.value
    plan.run_forward(st)
    plan.run_loss(st)
    plan.run_backward(st)
    y._adam(st)
    assert hip.hipStreamEndCapture(stream, ctypes.byref(graph)) == 0
dot = os.path.abspath(prefix + '.dot')
rc = hip.hipGraphDebugDotPrint(graph, dot.encode(), 1)       # hipGraphDebugDotFlagsVerbose
print('hipGraphDebugDotPrint rc', rc)
txt = open(dot).read()
nodes = dict(re.findall(r'"?(\w+)"?\s*\[[^\]]*label="([^"]*)"', txt))
edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
kinds = {}
for nid, lab in nodes.items():
    k = 'memset' if 'MEMSET' in lab.upper() else ('memcpy' if 'MEMCPY' in lab.upper() else ('kernel' if 'KERNEL' in lab.upper() or 'kernel' in lab else 'other'))
    kinds.setdefault(k, []).append(nid)
print('graph %s: %d nodes, %d edges; by kind: %s' % (dot, len(nodes), len(edges), {k: len(v) for k, v in kinds.items()}))
for k in ('memset', 'memcpy', 'other'):
    for nid in kinds.get(k, []):
        ins = [nodes.get(a, a)[:60] for a, b in edges if b == nid]
        outs = [nodes.get(b, b)[:60] for a, b in edges if a == nid]
        print('  %s node %s "%s": %d in-edges %s, %d out-edges %s' % (k, nid, nodes[nid][:80].replace('\n', ' '), len(ins), ins, len(outs), outs))
