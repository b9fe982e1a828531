#!/usr/bin/env python3
"""build_lmdb.py -- folder of images + per-image CSV boxes -> train / test lmdb (reference: build_lmdb.py:46-160).

Same flags, key format ("<n>_<csv basename>:<sorted unique class ids>") and protobuf value
(isg_ai.ImageYoloBoxesPair) as the reference; the LMDB file is written by yolo3.lmdbio (no liblmdb here)."""
import argparse
import os
import random
import shutil

import numpy as np

from yolo3 import bbox_utils, imagereader, lmdbio
from yolo3.isg_ai_pb import ImageYoloBoxesPair


def record_key(boxes, txn_nb, name):
    """"<n>_<name>:<sorted unique class ids>" (build_lmdb.py:88-96)."""
    boxes = np.asarray(boxes, dtype=np.int32).reshape(-1, 5)
    present = np.unique(boxes[:, 4]).astype(np.int32)
    return '{}_{}:{}'.format(txn_nb, name, ','.join(str(k) for k in present)).encode('ascii')


def encode_record(img, boxes):
    """Serialised isg_ai.ImageYoloBoxesPair of one example (build_lmdb.py:46-69)."""
    img = np.asarray(img, dtype=np.uint8)
    if img.ndim == 2:
        img = img[:, :, None]
    return ImageYoloBoxesPair.from_arrays(img, np.asarray(boxes, dtype=np.int32).reshape(-1, 5)).SerializeToString()


def make_record(img, boxes, txn_nb, name):
    """(key bytes, value bytes) of one example."""
    return record_key(boxes, txn_nb, name), encode_record(img, boxes)


def generate_database(csv_files, img_files, output_folder, database_name):
    print('Generating database {}'.format(database_name))
    out = os.path.join(output_folder, database_name)
    if os.path.exists(out):
        print('Deleting existing database')
        shutil.rmtree(out)
    # keys need the boxes only (cheap); images are read, serialised and written one at a time in key order
    # (the reference commits every 1000 records, build_lmdb.py:97-100 -- the dataset never sits in memory)
    source = {}
    for n, (img_fp, csv_fp) in enumerate(zip(img_files, csv_files)):
        name, _ = os.path.splitext(os.path.basename(csv_fp))
        boxes = bbox_utils.load_boxes_to_xywhc(csv_fp)
        source[record_key(boxes, n, name)] = (img_fp, boxes)

    def value_of(key):
        img_fp, boxes = source[key]
        return encode_record(imagereader.imread(img_fp), boxes)

    lmdbio.write_environment_stream(out, sorted(source), value_of)
    with open(os.path.join(out, 'annotation_list.csv'), 'w') as fh:
        for csv_fp in csv_files:
            fh.write('{}\n'.format(os.path.splitext(os.path.basename(csv_fp))[0]))


def build_lmdb(image_folder, csv_folder, output_folder, dataset_name, train_fraction, image_format):
    os.makedirs(output_folder, exist_ok=True)
    csv_files = [f for f in os.listdir(csv_folder) if f.endswith('.csv')]
    random.shuffle(csv_files)
    img_files = [os.path.join(image_folder, fn.replace('.csv', '.{}'.format(image_format))) for fn in csv_files]
    csv_files = [os.path.join(csv_folder, fn) for fn in csv_files]
    idx = int(train_fraction * len(csv_files))
    generate_database(csv_files[:idx], img_files[:idx], output_folder, 'train-' + dataset_name + '.lmdb')
    generate_database(csv_files[idx:], img_files[idx:], output_folder, 'test-' + dataset_name + '.lmdb')


if __name__ == '__main__':
    parser = argparse.ArgumentParser(prog='build_lmdb', description='Script which converts two folders of images and masks into a pair of lmdb databases for training.')
    parser.add_argument('--image_folder', dest='image_folder', type=str, required=True)
    parser.add_argument('--csv_folder', dest='csv_folder', type=str, required=True)
    parser.add_argument('--output_folder', dest='output_folder', type=str, required=True)
    parser.add_argument('--dataset_name', dest='dataset_name', type=str, required=True)
    parser.add_argument('--train_fraction', dest='train_fraction', type=float, default=0.8)
    parser.add_argument('--image_format', dest='image_format', type=str, default='tif')
    a = parser.parse_args()
    build_lmdb(a.image_folder, a.csv_folder, a.output_folder, a.dataset_name, a.train_fraction, a.image_format)
