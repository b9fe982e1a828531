"""Host mirror of the reference's bbox_utils.py (NMS part on the GPU).

per_class_nms / filter_small_boxes keep the reference signatures
(bbox_utils.py:240-281) but run the class-wise NMS kernel (csrc/detect.hip);
``detect`` is the fused entry the CLIs use: clip -> small-box filter ->
class-wise NMS on device-resident ``[N, Nb, 5+K]`` rows, no host round trip
of the candidates.  CSV writers follow bbox_utils.py:47-62 and 284-300.
"""
import numpy as np
import torch

from ._hip import lib, check


class _NmsBuffers:
    def __init__(self, n, nb, k, device):
        self.key = (n, nb, k, str(device))
        self.keep_idx = torch.empty(n, k, nb, dtype=torch.int32, device=device)
        self.keep_cnt = torch.zeros(n, k, dtype=torch.int32, device=device)
        self.keep_score = torch.empty(n, k, nb, dtype=torch.float32, device=device)
        self.ws_bytes = int(lib.y3_nms_workspace_bytes(n, nb, k))
        self.ws = torch.empty(self.ws_bytes // 4 + 4, dtype=torch.float32, device=device)


_cache = {}
_copy_streams = {}


def nms_device(rows, min_box_size=0.0, iou_threshold=0.3, score_threshold=0.1, clip_wh=None, private_outputs=False):
    """rows: CUDA float32 [N, Nb, 5+K].  Returns (keep_idx[N,K,Nb] int32,
    keep_cnt[N,K] int32, keep_score[N,K,Nb]) device tensors; entries beyond
    keep_cnt are undefined.  The outputs are per-shape cached buffers that the next call overwrites unless
    private_outputs is set (the workspace is always shared: launches on one stream run in order)."""
    assert rows.is_cuda and rows.dtype == torch.float32 and rows.dim() == 3
    rows = rows.contiguous()
    n, nb, d = rows.shape
    k = d - 5
    st_obj = torch.cuda.current_stream(rows.device)
    key = (n, nb, k, str(rows.device), st_obj.cuda_stream)       # one workspace per stream: launches on a stream run in order
    buf = _cache.get(key)
    if buf is None:
        buf = _cache[key] = _NmsBuffers(n, nb, k, rows.device)
    if private_outputs:
        keep_idx, keep_cnt, keep_score = torch.empty_like(buf.keep_idx), torch.zeros_like(buf.keep_cnt), torch.empty_like(buf.keep_score)
    else:
        keep_idx, keep_cnt, keep_score = buf.keep_idx, buf.keep_cnt, buf.keep_score
    cw, chh = (float(clip_wh[0]), float(clip_wh[1])) if clip_wh is not None else (-1.0, -1.0)
    check(lib.y3_nms_per_class(rows.data_ptr(), n, nb, k, float(min_box_size), float(score_threshold), float(iou_threshold), cw, chh,
                               keep_idx.data_ptr(), keep_cnt.data_ptr(), keep_score.data_ptr(), nb, buf.ws.data_ptr(),
                               buf.ws_bytes, st_obj.cuda_stream), 'y3_nms_per_class')
    return keep_idx, keep_cnt, keep_score


def detect_async(rows, min_box_size, iou_threshold=0.3, score_threshold=0.1, clip_wh=None):
    """Enqueue clip -> small-box filter -> class-wise NMS for a batch and return a ``collect()`` callable; nothing
    synchronises until it is called, so the caller can queue the next batch's network first.  ``rows`` must stay
    untouched until then (pass a clone of a buffer that the next forward overwrites)."""
    keep_idx, keep_cnt, keep_score = nms_device(rows, min_box_size, iou_threshold, score_threshold, clip_wh, private_outputs=True)
    n, nb, d = rows.shape
    k = d - 5
    done = torch.cuda.Event()
    done.record(torch.cuda.current_stream(rows.device))

    def collect():
        # the copies run on a stream of their own behind the NMS's event: queued on the NMS's stream they would also
        # wait for whatever the caller has put on it since (the next batches of a tiled image)
        cs = _copy_streams.get(str(rows.device))
        if cs is None:
            cs = _copy_streams[str(rows.device)] = torch.cuda.Stream(device=rows.device)
        cs.wait_event(done)
        with torch.cuda.stream(cs):
            return _collect()

    def _collect():
        cnt = keep_cnt.cpu().numpy()                       # the only synchronisation point besides the final copies
        out = [(None, None, None, None)] * n
        total = int(cnt.sum())
        if total == 0:
            return out
        # flat (image, class, slot) positions of every kept entry, class-major inside an image like bbox_utils.py:252-263
        ii = np.repeat(np.arange(n), cnt.sum(1))
        cc = np.concatenate([np.repeat(np.arange(k), cnt[i]) for i in range(n)])
        jj = np.concatenate([np.arange(c) for c in cnt.reshape(-1)])
        lin = torch.from_numpy((ii * k + cc) * nb + jj).to(rows.device)
        idx = keep_idx.view(-1)[lin].long()
        boxes = rows[torch.from_numpy(ii).to(rows.device), idx, 0:4]
        if clip_wh is not None:
            boxes[:, 0::2] = boxes[:, 0::2].clamp(0, float(clip_wh[0]))
            boxes[:, 1::2] = boxes[:, 1::2].clamp(0, float(clip_wh[1]))
        boxes, sc, idx = boxes.cpu().numpy(), keep_score.view(-1)[lin].cpu().numpy(), idx.cpu().numpy().astype(np.int32)
        lab = cc.astype('int32')
        ends = np.cumsum(cnt.sum(1))
        for i in range(n):
            a, b = int(ends[i] - cnt[i].sum()), int(ends[i])
            if b > a:
                out[i] = (boxes[a:b], sc[a:b], lab[a:b], idx[a:b])
        return out
    return collect


def detect(rows, min_box_size, iou_threshold=0.3, score_threshold=0.1, clip_wh=None):
    """inference.py:62-79 for a batch: returns, per image, (boxes[M,4], score[M],
    label[M] int32, keep[M] row indices) as NumPy arrays, or (None,)*4."""
    return detect_async(rows, min_box_size, iou_threshold, score_threshold, clip_wh)()


def per_class_nms(boxes, objectness, class_probs, iou_threshold=0.3, score_threshold=0.1):
    """bbox_utils.py:240-271 (same signature and (None, None, None) convention)."""
    boxes = np.asarray(boxes, np.float32)
    rows = np.concatenate([boxes, np.asarray(objectness, np.float32).reshape(-1, 1), np.asarray(class_probs, np.float32)], axis=1)
    if rows.shape[0] == 0:
        return None, None, None
    r = torch.from_numpy(rows).cuda()[None]
    b, s, l, _ = detect(r, -np.inf, iou_threshold, score_threshold)[0]
    return b, s, l


def single_class_nms(boxes, scores, iou_threshold):
    """bbox_utils.py:217-237: keep indices (selection order) into ``boxes``."""
    boxes = np.asarray(boxes, np.float32)
    scores = np.asarray(scores, np.float32).reshape(-1, 1)
    m = boxes.shape[0]
    if m == 0:
        return []
    rows = torch.from_numpy(np.ascontiguousarray(np.concatenate([boxes, scores], axis=1))).cuda()
    keep_idx = torch.empty(m, dtype=torch.int32, device=rows.device)
    keep_cnt = torch.zeros(1, dtype=torch.int32, device=rows.device)
    keep_score = torch.empty(m, dtype=torch.float32, device=rows.device)
    ws_bytes = int(lib.y3_nms_workspace_bytes(1, m, 1))
    ws = torch.empty(ws_bytes // 4 + 4, dtype=torch.float32, device=rows.device)
    st = torch.cuda.current_stream(rows.device).cuda_stream
    check(lib.y3_nms_single_class(rows.data_ptr(), m, float(iou_threshold), keep_idx.data_ptr(), keep_cnt.data_ptr(), keep_score.data_ptr(),
                                  ws.data_ptr(), ws_bytes, st), 'y3_nms_single_class')
    return [int(v) for v in keep_idx[:int(keep_cnt.item())].cpu().numpy()]


def compute_iou(box, boxes, box_area=None, boxes_area=None):
    """bbox_utils.py:200-214: IoU of one corner box [4] against boxes [M,4] (no +1) on the GPU, float32 in the
    reference's operation order -> ndarray [M].  box_area / boxes_area are accepted for signature compatibility; the
    reference only ever passes the areas of these same boxes (bbox_utils.py:232), which the kernel recomputes."""
    boxes = np.ascontiguousarray(np.asarray(boxes, np.float32).reshape(-1, 4))
    if boxes.shape[0] == 0:
        return np.zeros((0,), np.float32)
    b = torch.from_numpy(np.ascontiguousarray(np.asarray(box, np.float32).reshape(4))).cuda()
    bb = torch.from_numpy(boxes).cuda()
    out = torch.empty(boxes.shape[0], dtype=torch.float32, device=bb.device)
    check(lib.y3_compute_iou(b.data_ptr(), bb.data_ptr(), boxes.shape[0], 4, out.data_ptr(), torch.cuda.current_stream(bb.device).cuda_stream),
          'y3_compute_iou')
    return out.cpu().numpy()


def filter_small_boxes(boxes, min_size):
    """bbox_utils.py:274-281: rows [M, >=4] = x0, y0, x1, y1, ... -> the rows with width AND height strictly greater
    than min_size, original order.  The selection runs on the GPU (y3_filter_small_boxes); the CLIs use the path fused
    into the NMS launch (``detect``) instead."""
    boxes = np.asarray(boxes)
    if boxes.shape[0] == 0:
        return boxes
    rows = torch.from_numpy(np.ascontiguousarray(boxes[:, :4], dtype=np.float32)).cuda()
    idx = torch.empty(rows.shape[0], dtype=torch.int32, device=rows.device)
    cnt = torch.zeros(1, dtype=torch.int32, device=rows.device)
    check(lib.y3_filter_small_boxes(rows.data_ptr(), rows.shape[0], 4, float(min_size), idx.data_ptr(), cnt.data_ptr(),
                                    torch.cuda.current_stream(rows.device).cuda_stream), 'y3_filter_small_boxes')
    return boxes[idx[:int(cnt.item())].cpu().numpy().astype(np.int64), :]


def load_boxes_to_xywhc(filepath):
    """bbox_utils.py:106-124: CSV with header X,Y,W,H,C (skipinitialspace) -> float [n,5]; missing file -> [0,5]."""
    import csv
    import os
    rows = []
    if os.path.exists(filepath):
        with open(filepath) as fh:
            for row in csv.DictReader(fh, skipinitialspace=True):
                rows.append([int(row['X']), int(row['Y']), int(row['W']), int(row['H']), int(row['C'])])
    return np.asarray(rows, dtype=np.float64).reshape(-1, 5)


def load_boxes_to_ltrbc(filepath):
    """bbox_utils.py:83-103: as above with W,H converted to inclusive right / bottom."""
    a = load_boxes_to_xywhc(filepath)
    a[:, 2] = a[:, 0] + a[:, 2] - 1
    a[:, 3] = a[:, 1] + a[:, 3] - 1
    return a


def write_boxes_from_xywhc(boxes, csv_filename):
    """bbox_utils.py:47-62."""
    with open(csv_filename, 'w') as fh:
        fh.write('X,Y,W,H,C\n')
        for k in range(boxes.shape[0]):
            fh.write('{:d},{:d},{:d},{:d},{:d}\n'.format(int(boxes[k, 0]), int(boxes[k, 1]), int(boxes[k, 2]), int(boxes[k, 3]), int(boxes[k, 4])))


def write_boxes_from_ltrbc(boxes, csv_filename):
    """bbox_utils.py:65-80: [left, top, right, bottom, class] -> X,Y,W,H,C with W = right - left + 1."""
    with open(csv_filename, 'w') as fh:
        fh.write('X,Y,W,H,C\n')
        for k in range(boxes.shape[0]):
            x, y = boxes[k, 0], boxes[k, 1]
            fh.write('{:d},{:d},{:d},{:d},{:d}\n'.format(x, y, boxes[k, 2] - x + 1, boxes[k, 3] - y + 1, boxes[k, 4]))


def write_boxes_from_ltrbpc(boxes, csv_filename):
    """bbox_utils.py:284-300: W = x2 - x + 1."""
    with open(csv_filename, 'w') as fh:
        fh.write('X,Y,W,H,P,C\n')
        for k in range(boxes.shape[0]):
            x = int(boxes[k, 0])
            y = int(boxes[k, 1])
            w = int(boxes[k, 2] - x + 1)
            h = int(boxes[k, 3] - y + 1)
            fh.write('{:d},{:d},{:d},{:d},{:f},{:d}\n'.format(x, y, w, h, boxes[k, 4], int(boxes[k, 5])))
