"""yolo3: MI355X-native YOLOv3 train / inference hot path.

Host-side mirror of the reference's Python modules (model, bbox_utils,
imagereader, inference_tiled) over hand-written HIP kernels
(csrc/ -> _lib/libyolo3hip.so, C ABI in include/yolo3hip.h).
Importing the compute modules requires the built library; there is no CPU
fallback.
"""
__version__ = '0.1.0'
