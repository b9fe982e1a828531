"""Oracle (test infrastructure): NumPy restatement of the reference's host-side
post-processing.  Pinned by tests/golden/nms_*.npz (made by importing the
reference's bbox_utils.py; see tests/golden/make_golden.py).

Follows /root/reference/bbox_utils.py:
  compute_iou         200-214
  single_class_nms    217-237
  per_class_nms       240-271
  filter_small_boxes  274-281
All arithmetic is float32 elementwise in the reference's operation order, so
the GPU kernel can be compared bit-for-bit (keep indices) on tie-free scores.
"""
import numpy as np


def compute_iou(box, boxes, box_area=None, boxes_area=None):
    """bbox_utils.py:200-214 -- IoU of one corner box against many (no +1)."""
    x_left = np.maximum(box[0], boxes[:, 0])
    y_top = np.maximum(box[1], boxes[:, 1])
    x_right = np.minimum(box[2], boxes[:, 2])
    y_bottom = np.minimum(box[3], boxes[:, 3])
    inter = np.maximum(y_bottom - y_top, 0) * np.maximum(x_right - x_left, 0)
    if box_area is None:
        box_area = (box[2] - box[0]) * (box[3] - box[1])
    if boxes_area is None:
        boxes_area = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    unions = box_area + boxes_area - inter
    with np.errstate(divide='ignore', invalid='ignore'):
        return inter / unions


def single_class_nms(boxes, scores, iou_threshold):
    """bbox_utils.py:217-237 -- greedy NMS, keeps iou <= thr.  Returns indices
    into ``boxes`` in selection (descending-score) order."""
    areas = (boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1])
    order = scores.argsort()[::-1]
    keep = []
    thr = np.float32(iou_threshold)
    while order.size > 0:
        i = order[0]
        keep.append(int(i))
        order = order[1:]
        iou = compute_iou(boxes[i, :], boxes[order, :], areas[i], areas[order])
        order = order[np.where(iou <= thr)[0]]
    return keep


def filter_small_boxes_mask(rows, min_size):
    """bbox_utils.py:274-281 -- strict '>' on both extents."""
    w = rows[:, 2] - rows[:, 0]
    h = rows[:, 3] - rows[:, 1]
    return np.logical_and(w > min_size, h > min_size)


def filter_small_boxes(rows, min_size):
    return rows[filter_small_boxes_mask(rows, min_size), :]


def per_class_nms(boxes, objectness, class_probs, iou_threshold=0.3, score_threshold=0.1):
    """bbox_utils.py:240-271.  Returns (boxes, score, label) or (None,)*3."""
    num_classes = class_probs.shape[1]
    scores = np.sqrt(class_probs * objectness)
    pb, ps, pl = [], [], []
    for i in range(num_classes):
        idx = np.where(scores[:, i] >= np.float32(score_threshold))
        fb = boxes[idx]
        fs = scores[:, i][idx]
        if len(fb) == 0:
            continue
        k = single_class_nms(fb, fs, iou_threshold)
        pb.append(fb[k])
        ps.append(fs[k])
        pl.append(np.ones(len(k), dtype='int32') * i)
    if len(pb) == 0:
        return None, None, None
    return np.concatenate(pb, 0), np.concatenate(ps, 0), np.concatenate(pl, 0)


def detect_rows(rows, min_box_size, iou_threshold=0.3, score_threshold=0.1):
    """filter_small_boxes -> per_class_nms on raw model rows [Nb, 5+K], as
    inference.py:72-79 chains them; additionally returns, per class, the keep
    indices into the ORIGINAL row array (what the GPU kernel emits)."""
    rows = np.asarray(rows, dtype=np.float32)
    mask = filter_small_boxes_mask(rows, min_box_size)
    orig = np.where(mask)[0]
    r = rows[mask]
    K = rows.shape[1] - 5
    scores = np.sqrt(r[:, 5:] * r[:, 4:5])
    keep_per_class = []
    for c in range(K):
        cand = np.where(scores[:, c] >= np.float32(score_threshold))[0]
        if cand.size == 0:
            keep_per_class.append(np.zeros((0,), np.int32))
            continue
        k = single_class_nms(r[cand, 0:4], scores[cand, c], iou_threshold)
        keep_per_class.append(orig[cand[k]].astype(np.int32))
    return keep_per_class
