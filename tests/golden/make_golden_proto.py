#!/usr/bin/env python3
"""Golden protobuf bytes from the reference's own generated module (isg_ai_pb2.py).
Run in the build container only:
    cd /tmp && PROTOCOL_BUFFERS_PYTHON_IMPLEMENTATION=python PYTHONDONTWRITEBYTECODE=1 python3 /root/repo/tests/golden/make_golden_proto.py
(the generated code predates protobuf 3.19 and needs the pure-Python runtime)."""
import os
import sys

import numpy as np

sys.path.insert(0, '/root/reference')
from isg_ai_pb2 import ImageYoloBoxesPair   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(70)
img = rng.integers(0, 256, (6, 5, 3), dtype=np.uint8)
boxes = np.array([[1, 2, 3, 4, 0], [0, 0, 5, 6, 1]], dtype=np.int32)
# build_lmdb.write_img_to_db (build_lmdb.py:46-69) field by field
m = ImageYoloBoxesPair()
m.channels = 3
m.img_height, m.img_width = 6, 5
m.image = img.tobytes()
m.box_count = 2
m.boxes = boxes.tobytes()
m.img_type = img.dtype.str
m.box_type = boxes.dtype.str
a = m.SerializeToString()
e = ImageYoloBoxesPair()
img1 = rng.integers(0, 65535, (4, 4, 1)).astype(np.uint16)
e.channels = 1
e.img_height, e.img_width = 4, 4
e.image = img1.tobytes()
e.box_count = 0
e.img_type = img1.dtype.str
e.box_type = np.dtype(np.int32).str
b = e.SerializeToString()
np.savez(os.path.join(HERE, 'proto_pair.npz'), img=img, boxes=boxes, msg=np.frombuffer(a, np.uint8), img1=img1, msg1=np.frombuffer(b, np.uint8))
print(len(a), len(b))
