"""Host mirror of the reference's imagereader.py pieces that sit on the hot-path
contract: z-score normalisation (GPU kernel), HWC->CHW formatting and the
ground-truth label layout the loss consumes.

zscore_normalize  imagereader.py:34-46   (csrc/pointwise.hip, fp64 partial sums)
format_image      imagereader.py:57-60
format_boxes      ImageReader.__format_boxes, imagereader.py:252-324 (host NumPy,
                  as in the reference: it runs in the reader processes)
"""
import numpy as np
import torch

from ._hip import lib, check

NETWORK_DOWNSAMPLE_FACTOR = 32   # model.YoloV3.NETWORK_DOWNSAMPLE_FACTOR (model.py:25)


def zscore_normalize_device(x):
    """x: CUDA float32 tensor [B, ...]; each x[b] is normalised with its own
    whole-tensor mean / population std (imagereader.py:34-46)."""
    assert x.is_cuda and x.dtype == torch.float32
    x = x.contiguous()
    b = x.shape[0]
    count = x[0].numel()
    out = torch.empty_like(x)
    ws = torch.empty(int(lib.y3_zscore_workspace_bytes(b)) // 8 + 1, dtype=torch.float64, device=x.device)
    st = torch.cuda.current_stream(x.device).cuda_stream
    check(lib.y3_zscore(x.data_ptr(), out.data_ptr(), b, count, ws.data_ptr(), st), 'y3_zscore')
    return out


def zscore_normalize(image_data):
    """imagereader.py:34-46 on one image (any layout / dtype) -> float32 ndarray."""
    x = torch.from_numpy(np.ascontiguousarray(np.asarray(image_data).astype(np.float32))).cuda()
    return zscore_normalize_device(x[None])[0].cpu().numpy()


def format_image(image_data):
    """imagereader.py:57-60: HWC -> CHW."""
    return np.transpose(image_data, [2, 0, 1])


def format_boxes(boxes, image_size, anchors, number_classes):
    """ImageReader.__format_boxes (imagereader.py:252-324).

    boxes [n,5] = [x, y, w, h, class] with (x, y) the top-left corner.  Returns
    three float32 label tensors [G, G, A, 5+K] (G = size/32, /16, /8).  The box
    centre is floor(xy + (wh-1)/2); the best anchor (IoU of co-centred w,h) is
    written at the SAME anchor slot of all three scales (Q5)."""
    anchors = np.asarray(anchors, dtype=np.float32)
    num_anchors = len(anchors)
    f = NETWORK_DOWNSAMPLE_FACTOR
    grid_sizes = [(int(image_size[0] / f), int(image_size[1] / f)),
                  (int(image_size[0] / (f / 2)), int(image_size[1] / (f / 2))),
                  (int(image_size[0] / (f / 4)), int(image_size[1] / (f / 4)))]
    label = [np.zeros((g[0], g[1], num_anchors, 5 + number_classes), dtype=np.float32) for g in grid_sizes]
    if boxes is None or boxes.shape[0] == 0:
        return label
    boxes = boxes.astype(np.float32)
    box_wh = boxes[:, 2:4]
    boxes[:, 0:2] = np.floor(boxes[:, 0:2] + ((box_wh - 1) / 2.0))
    wh = np.expand_dims(box_wh, -2)
    inter_wh = np.maximum(np.minimum(wh / 2.0, anchors / 2.0) - np.maximum(-wh / 2.0, -anchors / 2.0), 0.0)
    inter = inter_wh[..., 0] * inter_wh[..., 1]
    iou = inter / (wh[..., 0] * wh[..., 1] + anchors[:, 0] * anchors[:, 1] - inter)
    best_anchor = np.argmax(iou, axis=-1)
    for t, n in enumerate(best_anchor):
        for l, g in enumerate(grid_sizes):
            i = np.floor(boxes[t, 1] / image_size[0] * g[0]).astype('int32')
            j = np.floor(boxes[t, 0] / image_size[1] * g[1]).astype('int32')
            c = boxes[t, 4].astype('int32')
            label[l][i, j, n, 0:4] = boxes[t, 0:4]
            label[l][i, j, n, 4] = 1.0
            label[l][i, j, n, 5 + c] = 1.0
    return label
