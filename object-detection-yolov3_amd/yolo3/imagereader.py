"""Host mirror of the reference's imagereader.py pieces that sit on the hot-path
contract: z-score normalisation (GPU kernel), HWC->CHW formatting and the
ground-truth label layout the loss consumes.

ImageReader       imagereader.py:79-460  (lmdb + protobuf datasets, worker processes, bounded queue)
zscore_normalize  imagereader.py:34-46   (csrc/pointwise.hip, fp64 partial sums)
format_image      imagereader.py:57-60
format_boxes      ImageReader.__format_boxes, imagereader.py:252-324 (host NumPy,
                  as in the reference: it runs in the reader processes)
"""
import multiprocessing
import os
import queue
import random
import traceback

import numpy as np
import torch

from ._hip import lib, check
from . import augment
from . import lmdbio
from .isg_ai_pb import ImageYoloBoxesPair

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


def inverse_format_boxes(label, batch_id):
    """imagereader.py:63-76: boxes [n,4] = x, y, w, h (top-left) back from a label tensor [B,G,G,A,5+K], anchor 0 only,
    in np.nonzero order; modifies the label rows in place like the reference does."""
    boxes = []
    ii, jj = np.nonzero(label[batch_id, :, :, 0, 4])
    for k in range(len(ii)):
        bb = label[batch_id, ii[k], jj[k], 0, 0:4]
        bb[0] = bb[0] - int(bb[2] / 2)
        bb[1] = bb[1] - int(bb[3] / 2)
        boxes.append(bb)
    return np.vstack(boxes)


def imread(fp):
    """imagereader.py:49-50 (skimage.io.imread there; PIL here).  Returns HWC or HW ndarray."""
    from PIL import Image
    return np.asarray(Image.open(fp))


def imwrite(img, fp):
    from PIL import Image
    Image.fromarray(np.asarray(img)).save(fp)


def _loader_process(reader, worker_id):
    reader._image_loader(worker_id)


class Dataset:
    """What get_tf_dataset() hands out: an iterable of (image[C,H,W], label_1, label_2, label_3); ``batch(n)`` stacks
    n examples and moves them to the GPU, where the images are z-scored by the HIP kernel (the reference z-scores in
    the reader processes on the CPU, imagereader.py:398); ``prefetch(d)`` assembles up to d batches ahead in pinned host
    memory on a background thread (tf.data's prefetch, train.py:61), so that taking examples off the worker queue and
    stacking them overlaps the GPU step instead of preceding it."""

    def __init__(self, reader, batch_size=None, device=None, prefetch_depth=0):
        self.reader, self.batch_size, self.device, self.prefetch_depth = reader, batch_size, device, prefetch_depth

    def batch(self, n):
        return Dataset(self.reader, int(n), self.device, self.prefetch_depth)

    def prefetch(self, n):
        return Dataset(self.reader, self.batch_size, self.device, max(1, min(int(n), 4)))     # batches, not examples: 4 is plenty

    def shard(self, num_shards, index):
        """experimental_distribute_dataset (train.py:62,66): every replica reads its own examples.  The reader's
        workers are forked at startup(), so the shard must be fixed before that (ImageReader(..., num_shards, shard_index));
        on a started reader this only checks that the request matches."""
        self.reader.set_shard(num_shards, index)
        return self

    def _examples(self):
        """Lists of batch_size examples off the worker queue; a short tail is dropped like tf.data drop_remainder."""
        gen = self.reader.raw_generator()
        while True:
            ex = []
            for e in gen:
                ex.append(e)
                if len(ex) == self.batch_size:
                    break
            if len(ex) < self.batch_size:
                return
            yield ex

    def __iter__(self):
        if self.batch_size is None:
            yield from self.reader.generator()           # z-scored examples, as the reference's unbatched dataset yields them
            return
        dev = self.device or torch.device('cuda', torch.cuda.current_device())
        if not self.prefetch_depth:
            for ex in self._examples():
                imgs = torch.from_numpy(np.stack([e[0] for e in ex])).to(dev, non_blocking=True)
                labels = [torch.from_numpy(np.stack([e[i] for e in ex])).to(dev, non_blocking=True) for i in (1, 2, 3)]
                yield (zscore_normalize_device(imgs), *labels)
            return
        import queue
        import threading
        ready = queue.Queue(maxsize=self.prefetch_depth)
        ring = [dict(bufs=None, event=None) for _ in range(self.prefetch_depth + 2)]
        stop = threading.Event()

        def hand_over(item):
            while not stop.is_set():
                try:
                    ready.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        def producer():
            try:
                for i, ex in enumerate(self._examples()):
                    if stop.is_set():
                        return
                    slot = ring[i % len(ring)]
                    if slot['event'] is not None:
                        slot['event'].synchronize()          # the upload that last used these pinned buffers has finished
                    if slot['bufs'] is None:
                        slot['bufs'] = [torch.empty((len(ex),) + ex[0][j].shape, dtype=torch.from_numpy(ex[0][j]).dtype, pin_memory=True)
                                        for j in range(4)]
                    for j in range(4):
                        np.stack([e[j] for e in ex], out=slot['bufs'][j].numpy())
                    if not hand_over(slot):
                        return
            finally:
                hand_over(None)

        threading.Thread(target=producer, name='yolo3-prefetch', daemon=True).start()
        try:
            while True:
                slot = ready.get()
                if slot is None:
                    return
                dev_t = [b.to(dev, non_blocking=True) for b in slot['bufs']]
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
                slot['event'] = ev
                yield (zscore_normalize_device(dev_t[0]), dev_t[1], dev_t[2], dev_t[3])
        finally:
            stop.set()          # the consumer walked away (epoch boundary): the producer exits at its next hand-over


class ImageReader:
    """imagereader.ImageReader (imagereader.py:79-460): same constructor, startup / shutdown / get_image_size /
    get_number_classes / get_image_count / get_example / generator; get_tf_dataset() returns a ``Dataset``."""

    def __init__(self, img_db, anchors, use_augmentation=True, balance_classes=False, shuffle=True, num_workers=1, num_shards=1, shard_index=0):
        """num_shards / shard_index (addition): with one process per GPU every rank owns a reader; an unshuffled reader
        (the test set) then serves every num_shards-th stride of the key list, so the ranks evaluate disjoint images like
        the replicas of the reference's one distributed test batch."""
        self.num_shards, self.shard_index = int(num_shards), int(shard_index)
        assert 0 <= self.shard_index < self.num_shards
        self.image_db = img_db
        self.use_augmentation = use_augmentation
        self.queue_starvation = False
        self.balance_classes = balance_classes
        self.anchors = anchors
        self.number_anchors = len(anchors)
        if not os.path.exists(self.image_db):
            print('Could not load database file: ')
            print(self.image_db)
            raise Exception("Missing Database")
        self.shuffle = shuffle
        random.seed()
        self.keys_flat = []
        self.keys = [[]]
        env = lmdbio.Environment(self.image_db)
        all_keys = list(env.keys())
        empty_images_flag = False
        highest = 0
        for key in all_keys:                                      # key = "<n>_<name>:<c0,c1,...>" (build_lmdb.py:90-96)
            for k in key.decode('ascii').split(':')[1].split(','):
                if len(k) == 0:
                    empty_images_flag = True
                else:
                    highest = max(highest, int(k))
        for _ in range(highest):
            self.keys.append([])
        if empty_images_flag:
            self.keys.append([])
        for key in all_keys:
            self.keys_flat.append(key)
            for k in key.decode('ascii').split(':')[1].split(','):
                idx = 0 if len(k) == 0 else (int(k) + 1 if empty_images_flag else int(k))
                self.keys[idx].append(key)
        datum = ImageYoloBoxesPair().ParseFromString(env.get(self.keys_flat[0]))
        self.image_size = [datum.img_height, datum.img_width, datum.channels]
        env.close()
        self.number_classes = len(self.keys) - 1 if empty_images_flag else len(self.keys)
        print('Found images of shape: {}'.format(self.image_size))
        print('Dataset has {} examples'.format(len(self.keys_flat)))
        self.nb_workers = num_workers
        self.maxOutQSize = num_workers * 10
        ctx = multiprocessing.get_context('fork')
        self._ctx = ctx
        self.terminateQ = ctx.Queue(maxsize=self.nb_workers)
        self.outQ = ctx.Queue(maxsize=self.maxOutQSize)
        self.workers = None
        self.done = False

    def get_image_size(self):
        return self.image_size

    def get_number_classes(self):
        return self.number_classes

    def get_image_count(self):
        return int(len(self.keys_flat))

    def set_shard(self, num_shards, index):
        if self.workers:
            if (int(num_shards), int(index)) != (self.num_shards, self.shard_index):
                raise RuntimeError('reader already started as shard {}/{}; pass num_shards / shard_index to ImageReader()'.format(
                    self.shard_index, self.num_shards))
            return
        assert 0 <= int(index) < int(num_shards)
        self.num_shards, self.shard_index = int(num_shards), int(index)

    def startup(self):
        self.done = False
        self.workers = [self._ctx.Process(target=_loader_process, args=(self, i), daemon=True) for i in range(self.nb_workers)]
        for w in self.workers:
            w.start()

    def shutdown(self):
        if not self.workers:
            return
        for _ in self.workers:
            self.terminateQ.put(None)
        got = 0
        while got < len(self.workers):                      # drain so blocked workers can finish (imagereader.py:203-222)
            try:
                while True:
                    if self.outQ.get(timeout=0.2) is None:
                        got += 1
            except queue.Empty:
                if not any(w.is_alive() for w in self.workers):
                    break
        for w in self.workers:
            w.join(5)
        self.workers = None

    def _next_key(self, state):
        if self.shuffle:
            if self.balance_classes:                            # imagereader.py:226-240
                while True:
                    label_idx = random.randint(0, len(self.keys) - 1)
                    if len(self.keys[label_idx]) > 0:
                        break
                return self.keys[label_idx][random.randint(0, len(self.keys[label_idx]) - 1)]
            return self.keys_flat[random.randint(0, len(self.keys_flat) - 1)]
        # no shuffle: stride the flat key list by worker id (Q17).  The reference indexes keys_flat[worker id] unguarded and
        # raises IndexError when a database has fewer images than reader processes; wrap instead.
        # With several shards (one reader per rank) worker w of shard s starts at s * nb_workers + w and strides by
        # num_shards * nb_workers: together the ranks walk the key list exactly like one reader with all the workers.
        state['idx'] %= len(self.keys_flat)
        fn = self.keys_flat[state['idx']]
        state['idx'] = (state['idx'] + self.nb_workers * self.num_shards) % len(self.keys_flat)
        return fn

    def load_example(self, key, env):
        """One example as the workers produce it: (image[C,H,W] float32 NOT yet z-scored, label_1, label_2, label_3)."""
        datum = ImageYoloBoxesPair().ParseFromString(env.get(key))
        img, boxes = datum.to_arrays()
        if list(img.shape) != list(self.image_size):
            raise RuntimeError("Encountered unexpected image shape from database. Expected {}. Found {}.".format(self.image_size, img.shape))
        boxes = boxes.copy()
        crop_to = [self.image_size[0], self.image_size[1]]
        if self.use_augmentation:                               # severities of imagereader.py:369-391
            img, boxes = augment.augment_image_box_pair(img.astype(np.float32), boxes, reflection_flag=True, rotation_flag=False, crop_to=crop_to,
                                                        noise_augmentation_severity=0.03, scale_augmentation_severity=0.1,
                                                        blur_augmentation_max_sigma=2, box_size_augmentation_severity=0.03,
                                                        box_location_jitter_severity=0.03)
        if img.shape[0] != self.image_size[0] or img.shape[1] != self.image_size[1]:
            img, boxes = augment.crop_to_size(img, boxes, crop_to)
        img = np.ascontiguousarray(format_image(img)).astype(np.float32)
        labels = format_boxes(boxes, self.image_size, self.anchors, self.number_classes)
        return (img, labels[0], labels[1], labels[2])

    def _image_loader(self, worker_id):
        state = {'idx': self.shard_index * self.nb_workers + worker_id}
        try:
            random.seed()
            np.random.seed((os.getpid() * 2654435761) % (2 ** 32))
            env = lmdbio.Environment(self.image_db)
            while True:
                try:
                    if self.terminateQ.get_nowait() is None:
                        break
                except queue.Empty:
                    pass
                self.outQ.put(self.load_example(self._next_key(state), env))
        except Exception as e:                                  # imagereader.py:413-417
            print('***************** Reader Error *****************')
            print(e)
            traceback.print_exc()
            print('***************** Reader Error *****************')
        finally:
            self.outQ.put(None)

    def _get_raw(self):
        if self.outQ.qsize() < int(0.1 * self.maxOutQSize):     # imagereader.py:424-430
            if not self.queue_starvation:
                print('Input Queue Starvation !!!!')
            self.queue_starvation = True
        if self.queue_starvation and self.outQ.qsize() > int(0.5 * self.maxOutQSize):
            print('Input Queue Starvation Over')
            self.queue_starvation = False
        return self.outQ.get()

    def get_example(self):
        """imagereader.py:420-436: one (image, label_1, label_2, label_3) with the image z-scored (imagereader.py:398 does it
        in the worker; here the workers hand out raw pixels because the batched path z-scores whole batches on the GPU, so the
        single-example accessors normalise on the way out -- on the GPU as well, there is no host z-score in this package)."""
        example = self._get_raw()
        if example is None:
            return None
        return (zscore_normalize(example[0]),) + tuple(example[1:])

    def generator(self):
        while True:
            example = self.get_example()
            if example is None:
                return
            yield example

    def raw_generator(self):
        """Examples as the workers produce them (image NOT z-scored): what Dataset.batch() stacks and normalises on the GPU."""
        while True:
            example = self._get_raw()
            if example is None:
                return
            yield example

    def get_queue_size(self):
        return self.outQ.qsize()

    def get_tf_dataset(self):
        """imagereader.py:443-460 (a torch-side iterable instead of tf.data)."""
        return Dataset(self)

    get_dataset = get_tf_dataset
