"""Data-parallel training: one process per GPU, RCCL all-reduce over xGMI.

Replaces ``tf.distribute.MirroredStrategy`` (train.py:38-39, model.py:510-515):
every replica runs its own forward/backward on its own 8 images; gradients are
SUMMED across replicas (no averaging -- the loss is already divided by the
global batch size, model.py:492 / SURVEY Q8) before the Keras-Adam update.

The gradient arena is laid out in layer-creation order and backward completes
layers from the last to the first, so the arena is cut into contiguous buckets
from its end; as soon as the wgrad of a bucket's lowest layer has been enqueued
the bucket's all-reduce is issued asynchronously (``torch.distributed`` runs it
on the process group's own stream, gated by an event on the compute stream) and
overlaps the remaining dgrad/wgrad kernels.  ``finish_step`` makes the compute
stream wait for all buckets before Adam reads the gradients.

Works with any ``torch.distributed`` backend: ``nccl`` (= RCCL) on GPUs,
``gloo`` in the CPU tests.
"""
import torch
import torch.distributed as dist


def make_buckets(layer_ranges, bucket_floats):
    """layer_ranges: [(lo, hi)] arena ranges (floats) of the layers in creation
    order, contiguous.  Returns [(lo, hi, first_layer)] from the END of the arena
    towards its start, each at least ``bucket_floats`` long (except the last)."""
    buckets = []
    hi = layer_ranges[-1][1]
    cur_lo = hi
    for i in range(len(layer_ranges) - 1, -1, -1):
        cur_lo = layer_ranges[i][0]
        if hi - cur_lo >= bucket_floats or i == 0:
            buckets.append((cur_lo, hi, i))
            hi = cur_lo
    return buckets


class DataParallel:
    def __init__(self, bucket_mb=32.0, group=None):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed must be initialised (backend nccl = RCCL on MI355X)')
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bucket_floats = int(bucket_mb * (1 << 20) / 4)
        self.num_replicas_in_sync = self.world_size      # MirroredStrategy attribute used at train.py:41
        self.model = None
        self._works = []

    def attach(self, model):
        """model needs ``grads`` (flat tensor) and ``specs`` with w_off / end_off."""
        self.model = model
        ranges = [(sp.w_off, sp.end_off) for sp in model.specs]
        self.buckets = make_buckets(ranges, self.bucket_floats)
        self._by_layer = {b[2]: b for b in self.buckets}
        model.dist = self
        return self

    def broadcast_parameters(self, *tensors):
        """Replicated variables start identical (MirroredStrategy semantics): rank 0's values win."""
        for t in tensors:
            dist.broadcast(t, src=0, group=self.group)

    # -- hooks called by YoloV3.train_step ---------------------------------------
    def begin_step(self):
        self._works = []

    def on_layer_done(self, layer_idx):
        b = self._by_layer.get(layer_idx)
        if b is not None and self.world_size > 1:
            lo, hi, _ = b
            self._works.append(dist.all_reduce(self.model.grads[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish_step(self):
        for w in self._works:
            w.wait()          # nccl: compute stream waits on the collective's stream; gloo: host wait
        self._works = []

    def reduce_sum(self, value):
        """strategy.reduce(SUM, per_replica_loss) (model.py:513)."""
        t = value.detach().clone() if torch.is_tensor(value) else torch.tensor(float(value))
        if self.world_size > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def mean_moving_stats(self, moving):
        """BN moving statistics are per-replica sync-on-read variables: reading them (a checkpoint save) aggregates by MEAN
        (App. C4) and leaves every replica's own running value untouched.  Returns the mean as a NEW tensor."""
        out = moving.detach().clone()
        if self.world_size > 1:
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
            out /= self.world_size
        return out
