#!/bin/bash
# Wall time of the real CLI on a synthetic lmdb: python train.py ... (2 epochs of 40 steps), prints the last lines of its log.
set -e
TMP=$(mktemp -d)
python - <<PY
import os, sys
sys.path.insert(0, 'object-detection-yolov3_amd')
import numpy as np, build_lmdb
from yolo3 import lmdbio
rng = np.random.default_rng(3)
for split, cnt in (('train', 64), ('test', 16)):
    items = []
    for i in range(cnt):
        img = rng.integers(0, 256, (416, 416, 3), dtype=np.uint8)
        k = int(rng.integers(1, 5))
        wh = rng.integers(40, 200, (k, 2))
        xy = np.stack([rng.integers(0, 416 - wh[:, 0]), rng.integers(0, 416 - wh[:, 1])], 1)
        items.append(build_lmdb.make_record(img, np.concatenate([xy, wh, rng.integers(0, 2, (k, 1))], 1).astype(np.int32), i, 'img%03d' % i))
    lmdbio.write_environment(os.path.join('$TMP', '%s-syn.lmdb' % split), items)
PY
START=$(date +%s%N)
python object-detection-yolov3_amd/train.py --batch_size 8 --test_every_n_steps 40 --train_database $TMP/train-syn.lmdb --test_database $TMP/test-syn.lmdb --output_dir $TMP/out --early_stopping 5 --max_epochs 2 --use_augmentation 1 > $TMP/log.txt 2>&1
END=$(date +%s%N)
grep 'Epoch took' $TMP/log.txt; tail -4 $TMP/log.txt
echo "wall $(( (END - START) / 1000000 )) ms for 2 epochs x (41 train + 2 test) steps, process start-up included"
