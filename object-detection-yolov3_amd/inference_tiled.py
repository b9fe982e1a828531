#!/usr/bin/env python3
"""inference_tiled.py -- large image -> overlapping tiles (96-px ghost border, reflect padding) -> network + NMS per
tile -> ghost-band rejection -> global merge -> X,Y,W,H,P,C csv.  Reference: inference_tiled.py:29-382 (same flags).

The tile geometry and the merge rules are the reference's, including Q12 (tiles that were reflect-padded on the
left / top report their clamped origin).  Unlike the reference (one tile per model call, BATCH_SIZE unused at :25) the
image is uploaded once, tiles are cut (reflect padding included) and z-scored on the GPU, go through the network in
batches and are NMS'ed in one launch per batch.  convert_image_to_tiles stays as the host restatement the device
tiler is tested against."""
import argparse
import os

import numpy as np
import torch

from yolo3 import bbox_utils, imagereader
from yolo3.model import YoloV3

BATCH_SIZE = 25                   # tiles per network launch (the reference declares 8 at :25 and never uses it); 25 = a 4k image in 4 launches
EDGE_EFFECT_RANGE = 96            # inference_tiled.py:26
NETWORK_DOWNSAMPLE_FACTOR = 32


def convert_image_to_tiles(img, tile_size):
    """inference_tiled.py:29-100 -> (tiles, x origins, y origins)."""
    height, width = img.shape[0], img.shape[1]
    radius = [EDGE_EFFECT_RANGE, EDGE_EFFECT_RANGE]
    assert tile_size[0] % NETWORK_DOWNSAMPLE_FACTOR == 0 and tile_size[1] % NETWORK_DOWNSAMPLE_FACTOR == 0
    if tile_size[0] >= height:
        radius[0] = 0
    if tile_size[1] >= width:
        radius[1] = 0
    zone = [tile_size[0] - 2 * radius[0], tile_size[1] - 2 * radius[1]]
    tiles, xs, ys = [], [], []
    for i in range(0, height, zone[0]):
        for j in range(0, width, zone[1]):
            x_st, y_st = j - radius[1], i - radius[0]
            x_end, y_end = j + zone[1] + radius[1], i + zone[0] + radius[0]
            pre_x, pre_y = max(-x_st, 0), max(-y_st, 0)
            x_st, y_st = max(x_st, 0), max(y_st, 0)
            post_x, post_y = max(x_end - width, 0), max(y_end - height, 0)
            x_end, y_end = min(x_end, width), min(y_end, height)
            tile = img[y_st:y_end, x_st:x_end]
            if pre_x or post_x or pre_y or post_y:
                tile = np.pad(tile, pad_width=((pre_y, post_y), (pre_x, post_x), (0, 0)), mode='reflect')
            xs.append(x_st)          # the CLAMPED origin (Q12)
            ys.append(y_st)
            tiles.append(tile)
    return tiles, xs, ys


def tile_table(height, width, tile_size):
    """The tile walk of convert_image_to_tiles as numbers only: rows {y0, ny, pre_y, x0, nx, pre_x} for y3_tile_gather
    plus the clamped origins (xs, ys) that the merge step uses (Q12)."""
    radius = [EDGE_EFFECT_RANGE, EDGE_EFFECT_RANGE]
    assert tile_size[0] % NETWORK_DOWNSAMPLE_FACTOR == 0 and tile_size[1] % NETWORK_DOWNSAMPLE_FACTOR == 0
    if tile_size[0] >= height:
        radius[0] = 0
    if tile_size[1] >= width:
        radius[1] = 0
    zone = [tile_size[0] - 2 * radius[0], tile_size[1] - 2 * radius[1]]
    rows, xs, ys = [], [], []
    for i in range(0, height, zone[0]):
        for j in range(0, width, zone[1]):
            x_st, y_st = j - radius[1], i - radius[0]
            x_end, y_end = min(j + zone[1] + radius[1], width), min(i + zone[0] + radius[0], height)
            pre_x, pre_y = max(-x_st, 0), max(-y_st, 0)
            x_st, y_st = max(x_st, 0), max(y_st, 0)
            rows.append([y_st, y_end - y_st, pre_y, x_st, x_end - x_st, pre_x])
            xs.append(x_st)
            ys.append(y_st)
    return np.asarray(rows, np.int32), xs, ys


_BANDS = os.environ.get('Y3_TILED_BANDS', '1') != '0'      # (development: 0 = the whole image before the first batch)
_GATHER_DTYPES = {np.dtype(np.uint8): 0, np.dtype(np.uint16): 1, np.dtype(np.float32): 2}


def tiles_to_device(img_dev, dtype_code, img_shape, table_dev, t0, count, tile_size):
    """Tiles t0 .. t0+count of the device-resident HWC image -> float32 [count, C, th, tw] (y3_tile_gather)."""
    from yolo3._hip import lib, check
    h, w, c = img_shape
    out = torch.empty(count, c, tile_size[0], tile_size[1], dtype=torch.float32, device=img_dev.device)
    st = torch.cuda.current_stream(img_dev.device).cuda_stream
    check(lib.y3_tile_gather(img_dev.data_ptr(), dtype_code, h, w, c, table_dev.data_ptr() + 24 * t0, count, tile_size[0], tile_size[1],
                             out.data_ptr(), st), 'y3_tile_gather')
    return out


def merge_tile_detections(boxes, scores, class_label, tile_x, tile_y, tile_size, img_size):
    """Ghost-band rejection + shift to global coordinates for one tile (inference_tiled.py:230-266).
    boxes float32 [M,4] x1,y1,x2,y2 in tile coordinates.  Returns (boxes, scores, labels) or None."""
    scores = scores.reshape((-1, 1))
    class_label = class_label.reshape((-1, 1))
    cx = (boxes[:, 2] + boxes[:, 0]) / 2.0
    cy = (boxes[:, 3] + boxes[:, 1]) / 2.0
    cxg, cyg = cx + tile_x, cy + tile_y
    E = EDGE_EFFECT_RANGE
    invalid = ((cyg > E) & (cy < E)) | ((cyg <= img_size[0] - E) & (cy >= tile_size[0] - E)) | \
              ((cxg > E) & (cx < E)) | ((cxg <= img_size[1] - E) & (cx >= tile_size[1] - E))
    if np.any(invalid):
        boxes, scores, class_label = boxes[~invalid, :], scores[~invalid], class_label[~invalid]
    if boxes.shape[0] == 0:
        return None
    boxes = boxes.copy()
    boxes[:, 0] += tile_x
    boxes[:, 2] += tile_x
    boxes[:, 1] += tile_y
    boxes[:, 3] += tile_y
    return boxes, scores, class_label


def finalize_predictions(boxes_list, scores_list, class_label_list, img_size):
    """inference_tiled.py:272-310: concat, round, drop centres outside the image, clamp -> float64 [M,6]."""
    if len(boxes_list) > 0:
        boxes = np.round(np.concatenate(boxes_list, axis=0)).astype(np.int32)
        scores = np.concatenate(scores_list, axis=0)
        class_label = np.concatenate(class_label_list, axis=0)
        cx = (boxes[:, 2] + boxes[:, 0]) / 2.0
        cy = (boxes[:, 3] + boxes[:, 1]) / 2.0
        invalid = (cx < 0) | (cx >= img_size[1]) | (cy < 0) | (cy >= img_size[0])
        if np.any(invalid):
            boxes, scores, class_label = boxes[~invalid, :], scores[~invalid], class_label[~invalid]
        for col, lim in ((0, img_size[1]), (1, img_size[0]), (2, img_size[1]), (3, img_size[0])):
            boxes[boxes[:, col] < 0, col] = 0
            boxes[boxes[:, col] >= lim, col] = lim - 1
    else:
        boxes, scores, class_label = np.zeros((0, 4)), np.zeros((0, 1)), np.zeros((0, 1))
    return np.concatenate((boxes, scores, class_label), axis=-1)


def plan_tile_batches(n_tiles, tile_size, compute_units=256, mem_budget_bytes=None):
    """How many tiles go into each network launch on the bf16 conv path.  The layers that carry the FLOPs run on 256 x 256
    output tiles, one workgroup per CU (csrc/conv_bf16.hip, conv_bf16_pp_kernel), so a launch costs whole ROUNDS of 256
    workgroups: a batch of 25 tiles of 608^2 is 565 workgroups on the /8 stage -- three rounds for 2.2 rounds of work.  The
    cost of a batch of B tiles is modelled as  sum over the /8, /16, /32 stages of  (3x3 layers of the stage) x (relative K
    of a workgroup) x ceil(row tiles x column tiles / CUs),  and the batch size that minimises the cost of all n_tiles
    (remainder batch included) is taken: 45 + 45 + 10 for the 100 tiles of a 4096^2 image at 608^2 (measured network rate
    on MI355X: 3 660 tiles/s at 25, 4 520 at 44).  Returns the list of batch sizes."""
    th, tw = int(tile_size[0]), int(tile_size[1])
    # A batch also has to FIT: the bf16 plan keeps every layer's output (about 110 bf16 values per input pixel, measured: 45 tiles
    # of 608^2 take 3.7 GB) -- cap the batch so that it stays within a quarter of `mem_budget_bytes` (the caller passes the free
    # device memory; None: no cap).  The round model itself only knows the 256-CU MI355X with the three-stage 11 / 11 / 7 layer
    # weights it was measured on: for another CU count the plain BATCH_SIZE is used.
    cap = 64
    if mem_budget_bytes is not None:
        per_tile = 110 * 2 * th * tw
        cap = max(1, min(cap, int(0.25 * mem_budget_bytes // per_tile)))
    if compute_units != 256:
        b = max(1, min(BATCH_SIZE, cap, n_tiles))
        return [b] * (n_tiles // b) + ([n_tiles % b] if n_tiles % b else [])

    def cost(b):
        c = 0
        for stride, col_tiles, weight in ((8, 1, 11 * 1), (16, 2, 11 * 2), (32, 4, 7 * 4)):
            rows = -(-(b * (th // stride) * (tw // stride)) // 256)
            c += weight * -(-(rows * col_tiles) // compute_units)
        return c

    best = None
    for b in range(min(4, cap), cap + 1):
        full, rem = divmod(n_tiles, b)
        total = full * cost(b) + (cost(rem) if rem else 0)
        if best is None or total < best[0]:
            best = (total, b)
    b = min(best[1], n_tiles) if n_tiles > 0 else 1
    out = [b] * (n_tiles // b)
    if n_tiles % b:
        out.append(n_tiles % b)
    return out


def upload_bands(table, sizes, height):
    """Rows of the image that must be on the device before batch i of the tile table can be gathered: the cumulative maximum of
    y0 + ny over the tiles of batches 0..i (tile_table rows are {y0, ny, pre_y, x0, nx, pre_x}: a tile reads only its crop, the
    reflected border comes from inside it); the last batch completes the image."""
    out, hi, t0 = [], 0, 0
    for i, nb in enumerate(sizes):
        if i + 1 == len(sizes):
            hi = int(height)
        elif nb > 0:
            hi = max(hi, int((table[t0:t0 + nb, 0] + table[t0:t0 + nb, 1]).max()))
        out.append(hi)
        t0 += nb
    return out


def inference_image_tiled(yolo_model, img, tile_size, min_roi_size, batch_size=None):
    """inference_tiled.py:185-310.  ``yolo_model(batch, training=False)`` maps CUDA float32 [B,C,h,w] (z-scored) to
    rows [B, Nb, 5+K] (CUDA tensor or ndarray).  batch_size None: BATCH_SIZE tiles per launch on the fp32 path,
    plan_tile_batches() on the bf16 path."""
    img_size = img.shape
    # the image goes to the GPU once, in its own dtype; cropping, reflect padding, astype(float32) and HWC -> CHW of
    # convert_image_to_tiles (inference_tiled.py:29-100,199-203) happen there, one launch per batch of tiles
    img = np.ascontiguousarray(img)
    if img.dtype not in _GATHER_DTYPES:
        img = img.astype(np.float32)
    code = _GATHER_DTYPES[img.dtype]
    img_host = torch.from_numpy(img.view(np.int16) if code == 1 else img)
    table, xs, ys = tile_table(img_size[0], img_size[1], tile_size)
    table_dev = torch.from_numpy(table).cuda()
    boxes_list, scores_list, class_label_list = [], [], []

    def merge(collect, b0):
        for k, (boxes, scores, class_label, _) in enumerate(collect()):
            if boxes is None:
                continue
            r = merge_tile_detections(boxes, scores, class_label, xs[b0 + k], ys[b0 + k], tile_size, img_size)
            if r is not None:
                boxes_list.append(r[0])
                scores_list.append(r[1])
                class_label_list.append(r[2])

    if batch_size is None:
        bf16 = getattr(getattr(yolo_model, '_y', None), 'inference_precision', 'fp32') == 'bf16'
        if bf16:
            dev = torch.cuda.current_device()
            sizes = plan_tile_batches(len(xs), tile_size, compute_units=torch.cuda.get_device_properties(dev).multi_processor_count,
                                      mem_budget_bytes=torch.cuda.mem_get_info(dev)[0] + torch.cuda.memory_reserved(dev))
        else:
            sizes = None
        batch_size = BATCH_SIZE
    else:
        sizes = None
    if sizes is None:
        sizes = [min(batch_size, len(xs) - b0) for b0 in range(0, len(xs), batch_size)]
    starts = [sum(sizes[:i]) for i in range(len(sizes))]
    # Batches alternate between two streams with their own activation buffers (model slots).  Everything is queued first;
    # detections are copied back and merged at the end.  The image goes up in BANDS: a batch only needs the rows its tiles
    # cover, so the network starts after the first band (0.45 of a 4096^2 image for the first 45 of 100 tiles) and the other
    # bands travel on a copy stream while it runs (a 50 MB image takes ~1 ms over PCIe).
    cur = torch.cuda.current_stream()
    slots = 2 if getattr(yolo_model, 'supports_slots', False) else 1
    slots = min(slots, int(os.environ.get('Y3_TILED_SLOTS', slots)))      # (development: 1 = batches one after the other)
    streams = [torch.cuda.Stream() for _ in range(slots)] if slots > 1 else [cur]
    copy_stream = torch.cuda.Stream() if len(sizes) > 1 else cur
    img_dev = torch.empty(img_host.shape, dtype=img_host.dtype, device='cuda')
    for s in streams + [copy_stream]:
        if s is not cur:
            s.wait_stream(cur)                                            # the allocations above
            img_dev.record_stream(s)
            table_dev.record_stream(s)
    queued = []
    rows_up = 0
    bands = upload_bands(table, sizes, img_size[0]) if _BANDS else [img_size[0]] * len(sizes)
    fused = hasattr(yolo_model, 'run_tiles') and not getattr(yolo_model, '_fm', False) and os.environ.get('Y3_TILED_FUSED', '1') != '0'
    for bi, (b0, nb) in enumerate(zip(starts, sizes)):
        need = bands[bi]
        if need > rows_up:
            with torch.cuda.stream(copy_stream):
                img_dev[rows_up:need].copy_(img_host[rows_up:need])      # pageable memory: returns when the band has arrived
            rows_up = need
        with torch.cuda.stream(streams[bi % slots]):
            if fused:
                # tiles -> z-scored network input in two passes over the image, then the network (yolo3.model.YoloV3.predict_tiles)
                rows = yolo_model.run_tiles(img_dev, code, img_size, table_dev.data_ptr() + 24 * b0, nb, tile_size=tile_size, slot=bi % slots)
            else:
                x = tiles_to_device(img_dev, code, img_size, table_dev, b0, nb, tile_size)
                x = imagereader.zscore_normalize_device(x)               # per TILE statistics (inference_tiled.py:205, Q12)
                rows = yolo_model(x, training=False, slot=bi % slots) if slots > 1 else yolo_model(x, training=False)
            rows = torch.as_tensor(rows, dtype=torch.float32).cuda().clone()   # the slot's output buffer is reused two batches later
            queued.append((bbox_utils.detect_async(rows, min_roi_size), b0))
    for item in queued:
        merge(*item)
    for s in streams:
        cur.wait_stream(s)
    predictions = finalize_predictions(boxes_list, scores_list, class_label_list, img_size)
    print('Found: {} rois'.format(predictions.shape[0]))
    return predictions


def inference_image_folder(image_folder, image_format, saved_model_filepath, output_folder, tile_size, min_roi_size, precision='fp32',
                           batch_size=None):
    if not os.path.exists(saved_model_filepath):
        raise RuntimeError('Missing saved_model_filepath File')
    if image_format.startswith('.'):
        image_format = image_format[1:]
    img_filepath_list = [os.path.join(image_folder, fn) for fn in os.listdir(image_folder) if fn.endswith('.{}'.format(image_format))]
    # replicas only (no collective): under `python -m torch.distributed.run --nproc-per-node N` every rank takes every N-th image
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world > 1:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
        img_filepath_list = sorted(img_filepath_list)[rank::world]
    path = os.path.join(saved_model_filepath, 'yolov3.npz') if os.path.isdir(saved_model_filepath) else saved_model_filepath
    yolo = YoloV3.from_file(path)
    yolo.inference_precision = precision          # 'bf16': bf16 MFMA convs, fp32 heads / decode / NMS (BASELINE config 5)
    if list(tile_size) != list(yolo.img_size[:2]):
        raise RuntimeError('tile size {} must equal the size the model was trained at {} (Q18)'.format(tile_size, yolo.img_size[:2]))
    yolo_model = yolo.get_keras_model()
    os.makedirs(output_folder, exist_ok=True)
    print('Starting inference of file list')
    for i, img_filepath in enumerate(img_filepath_list):
        _, file_name = os.path.split(img_filepath)
        print('{}/{} : {}'.format(i, len(img_filepath_list), file_name))
        img = imagereader.imread(img_filepath)
        if len(img.shape) == 2:
            img = np.expand_dims(img, -1)
        predictions = inference_image_tiled(yolo_model, img, tile_size, min_roi_size, batch_size)
        bbox_utils.write_boxes_from_ltrbpc(predictions, os.path.join(output_folder, file_name.replace(image_format, 'csv')))


if __name__ == '__main__':
    parser = argparse.ArgumentParser(prog='inference', description='Script to detect stars with the selected model')
    parser.add_argument('--saved-model-filepath', type=str, required=True)
    parser.add_argument('--output-folder', type=str, required=True)
    parser.add_argument('--image-folder', dest='image_folder', type=str, required=True)
    parser.add_argument('--image-format', dest='image_format', type=str, default='tif')
    parser.add_argument('--tile-height', type=int, default=512)
    parser.add_argument('--tile-width', type=int, default=512)
    parser.add_argument('--min-box-size', type=int, default=32)
    parser.add_argument('--precision', choices=['fp32', 'bf16'], default='fp32', help='conv arithmetic (extension; the reference is fp32)')
    parser.add_argument('--batch-size', type=int, default=None, help='tiles per network launch (extension; default: %d on the fp32 path, planned per image on the bf16 path)' % BATCH_SIZE)
    a = parser.parse_args()
    inference_image_folder(a.image_folder, a.image_format, a.saved_model_filepath, a.output_folder, [a.tile_height, a.tile_width], a.min_box_size, a.precision, a.batch_size)
