#!/usr/bin/env python3
"""inference.py -- per image: z-score -> network -> clip -> small-box filter -> class-wise NMS -> X,Y,W,H,C csv.
Reference: inference.py:24-135 (same flags).  --saved-model-filepath points at the weight file written by train.py
(<output_dir>/saved_model/yolov3.npz) instead of a TF SavedModel.  Everything between reading the image and writing
the csv runs on the GPU."""
import argparse
import os

import numpy as np
import torch

from yolo3 import bbox_utils, imagereader
from yolo3.model import YoloV3


def load_model(path):
    if os.path.isdir(path):
        path = os.path.join(path, 'yolov3.npz')
    return YoloV3.from_file(path)


def inference(image_folder, image_format, saved_model_filepath, output_folder, min_box_size, precision='fp32', batch_size=8):
    os.makedirs(output_folder, exist_ok=True)
    if image_format.startswith('.'):
        image_format = image_format[1:]
    img_filepath_list = [os.path.join(image_folder, fn) for fn in os.listdir(image_folder) if fn.endswith('.{}'.format(image_format))]
    # replicas only (no collective): under `python -m torch.distributed.run --nproc-per-node N` every rank takes every N-th image
    world, rank = int(os.environ.get('WORLD_SIZE', '1')), int(os.environ.get('RANK', '0'))
    if world > 1:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % torch.cuda.device_count())
        img_filepath_list = sorted(img_filepath_list)[rank::world]
    yolo = load_model(saved_model_filepath)
    yolo.inference_precision = precision          # 'bf16': bf16 MFMA convs, fp32 heads / decode / NMS (not in the reference)
    yolo_model = yolo.get_keras_model()
    print('Starting inference of file list')
    # The reference runs one image per model call (inference.py:40-101).  Here up to `batch_size` images go through the
    # network together; each is still z-scored with its own statistics and clipped / filtered / NMS'ed on its own.
    for g0 in range(0, len(img_filepath_list), batch_size):
        group = img_filepath_list[g0:g0 + batch_size]
        imgs = []
        for i, img_filepath in enumerate(group):
            print('{}/{} : {}'.format(g0 + i, len(img_filepath_list), os.path.split(img_filepath)[1]))
            img = imagereader.imread(img_filepath)
            imgs.append(img[:, :, None] if img.ndim == 2 else img)
        height, width, channels = imgs[0].shape
        if any(im.shape != imgs[0].shape for im in imgs):
            raise RuntimeError('images of one folder must share one size (the model input is fixed, Q18): {}'.format({im.shape for im in imgs}))
        x = torch.from_numpy(np.stack([np.ascontiguousarray(im.astype(np.float32).transpose((2, 0, 1))) for im in imgs])).cuda()
        x = imagereader.zscore_normalize_device(x)                       # per-image statistics (inference.py:49)
        rows = yolo_model(x, training=False)                              # [B, Nb, 5+K]
        # clip to the image (the intent of inference.py:62-65, Q11), small-box filter (:72), class-wise NMS (:79)
        dets = bbox_utils.detect(rows, min_box_size, clip_wh=(width, height))
        for img_filepath, (boxes, scores, class_label, _) in zip(group, dets):
            file_name = os.path.split(img_filepath)[1]
            if boxes is None:                                             # the reference would crash here (Q11); write an empty csv
                out = np.zeros((0, 5), np.int32)
            else:
                boxes[:, 2] = boxes[:, 2] - boxes[:, 0]
                boxes[:, 3] = boxes[:, 3] - boxes[:, 1]
                out = np.concatenate((boxes, class_label.reshape(-1, 1)), axis=-1).astype(np.int32)    # truncation (Q20)
            print('Found: {} rois'.format(out.shape[0]))
            bbox_utils.write_boxes_from_xywhc(out, os.path.join(output_folder, file_name.replace(image_format, 'csv')))


if __name__ == '__main__':
    parser = argparse.ArgumentParser(prog='inference', description='Script to detect stars with the selected model')
    parser.add_argument('--saved-model-filepath', type=str, help='Filepath to the saved model to use', required=True)
    parser.add_argument('--output-folder', type=str, required=True)
    parser.add_argument('--image-folder', dest='image_folder', type=str, required=True)
    parser.add_argument('--image-format', dest='image_format', type=str, default='tif')
    parser.add_argument('--min-box-size', type=int, default=32, help='Smallest detection to consider. Default (32, 32).')
    parser.add_argument('--precision', choices=['fp32', 'bf16'], default='fp32', help='conv arithmetic (extension; the reference is fp32)')
    parser.add_argument('--batch-size', type=int, default=8, help='images per model call (extension; the reference runs one)')
    a = parser.parse_args()
    print('Arguments:')
    for k, v in vars(a).items():
        print('{} = {}'.format(k, v))
    inference(a.image_folder, a.image_format, a.saved_model_filepath, a.output_folder, a.min_box_size, a.precision, a.batch_size)
