"""Data-parallel training: one process per GPU, RCCL all-reduce over xGMI.

Replaces ``tf.distribute.MirroredStrategy`` (train.py:38-39, model.py:510-515):
every replica runs its own forward/backward on its own 8 images; gradients are
SUMMED across replicas (no averaging -- the loss is already divided by the
global batch size, model.py:492 / SURVEY Q8) before the Keras-Adam update.

The gradient arena is laid out in layer-creation order and backward completes
layers from the last to the first, so the arena is cut into contiguous buckets
from its end; as soon as the wgrad of a bucket's lowest layer has been enqueued
the bucket's all-reduce is issued and overlaps the remaining dgrad/wgrad
kernels.  ``finish_step`` makes the compute stream wait for all buckets before
Adam reads the gradients.

Stream protocol: the caller has already joined the kernel-gradient side stream
into the compute stream (``_Plan._run``), so everything enqueued on the compute
stream at ``on_layer_done`` covers every gradient of the bucket.
  * torch transport on ``nccl``: the collective is issued with the compute stream
    current -- RCCL's own stream (inside the process group) waits for it -- and
    ``finish_step`` calls ``Work.wait()``, a device-side wait of the compute
    stream on RCCL's stream.  Three streams in all (compute, kernel gradients,
    RCCL): one more made the step 3 ms slower on MI355X (the runtime multiplexes
    streams onto 4 hardware queues; a collective's wait then sits in front of the
    kernel gradients that share its queue and the two-stream overlap is lost).
  * native transport, gloo, and any transport in ``collect_stats`` mode: the
    collective is issued on this object's own ``comm`` stream, which first waits
    for an event recorded on the compute stream; ``finish_step`` makes the
    compute stream wait for the ``comm`` stream's last event.
The host never blocks with RCCL.

Transports:
  * ``torch``  -- ``torch.distributed.all_reduce`` on the group's backend:
    ``nccl`` (= RCCL) on GPUs, ``gloo`` in the CPU tests / one-GPU rehearsals
    (gloo's ``Work.wait()`` blocks the host).
  * ``native`` -- RCCL called directly through the C ABI (``y3_comm_*``,
    csrc/comm.hip) on the ``comm`` stream; the torch process group (any
    backend) only bootstraps the 128-byte unique id and carries the scalar
    reductions.  Opt-in (``transport='native'`` / ``Y3_DP_TRANSPORT=native``).

``force_collective`` (``Y3_DP_FORCE=1``) issues the collectives even with one
rank, so that the whole stream / event / process-group machinery is exercised
on a single-GPU box (tests/test_gpu_dist.py).
"""
import ctypes
import os
import time

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
    def __init__(self, bucket_mb=32.0, group=None, force_collective=None, transport=None, collect_stats=False):
        if not dist.is_initialized():
            raise RuntimeError('torch.distributed must be initialised (backend nccl = RCCL on MI355X)')
        self.group = group
        self.world_size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)
        self.bucket_floats = int(bucket_mb * (1 << 20) / 4)
        self.num_replicas_in_sync = self.world_size      # MirroredStrategy attribute used at train.py:41
        if force_collective is None:
            force_collective = os.environ.get('Y3_DP_FORCE', '0') == '1'
        self.active = self.world_size > 1 or bool(force_collective)
        self.transport = transport or os.environ.get('Y3_DP_TRANSPORT', 'torch')
        if self.transport not in ('torch', 'native'):
            raise ValueError("transport must be 'torch' or 'native'")
        self.collect_stats = bool(collect_stats)
        self.time_wait = False      # record events around finish_step's wait WITHOUT changing the stream protocol (the timed, "lean" topology)
        self.model = None
        self.comm = None            # HIP stream the collectives are issued on
        self._native = None         # ncclComm_t of the native transport
        self._works = []
        self._last = None           # event on the comm stream behind the newest collective
        self._timing = []           # (start, end) events on the comm stream, stats mode
        self._wait_span = None      # (before, after) events on the compute stream around finish_step's wait
        self._host_wait_s = 0.0
        self.last_stats = None

    # -- setup ---------------------------------------------------------------------
    def attach(self, model):
        """model needs ``grads`` (flat tensor) and ``specs`` with w_off / end_off."""
        self.model = model
        ranges = [(sp.w_off, sp.end_off) for sp in model.specs]
        self.buckets = make_buckets(ranges, self.bucket_floats)
        self._by_layer = {b[2]: b for b in self.buckets}
        model.dist = self
        if model.grads.is_cuda:
            from . import streams
            self.comm = streams.reserve(model.grads.device)[1]       # bound to its hardware queue before any communicator existed
            if self.transport == 'native' and self._native is None:
                self._init_native(model.grads.device)
        elif self.transport == 'native':
            raise RuntimeError('the native RCCL transport needs device tensors')
        return self

    def _init_native(self, device):
        """ncclCommInitRank through the C ABI: rank 0 draws the unique id, the torch group carries its 128 bytes."""
        from ._hip import lib, check
        uid = (ctypes.c_ubyte * 128)()
        if self.rank == 0:
            check(lib.y3_comm_unique_id(uid), 'y3_comm_unique_id')
        t = torch.tensor(list(bytes(uid)), dtype=torch.uint8)
        if self.backend == 'nccl':
            t = t.to(device)
        if self.world_size > 1:
            dist.broadcast(t, src=0, group=self.group)
        uid = (ctypes.c_ubyte * 128)(*t.cpu().tolist())
        comm = ctypes.c_void_p()
        with torch.cuda.device(device):
            check(lib.y3_comm_init(uid, self.world_size, self.rank, ctypes.byref(comm)), 'y3_comm_init')
        self._native = comm

    def close(self):
        if self._native is not None:
            from ._hip import lib, check
            torch.cuda.synchronize()
            check(lib.y3_comm_destroy(self._native), 'y3_comm_destroy')
            self._native = None

    def comm_info(self):
        """What the communicator itself reports (goes into bench.py's ``comm`` object)."""
        info = dict(backend=self.backend, transport=self.transport, world_size=self.world_size, buckets=len(getattr(self, 'buckets', [])),
                    bucket_mb=self.bucket_floats * 4 / float(1 << 20), forced=self.active and self.world_size == 1)
        if self._native is not None:
            from ._hip import lib, check
            n, ver = ctypes.c_int(-1), ctypes.c_int(-1)
            check(lib.y3_comm_info(self._native, ctypes.byref(n), ctypes.byref(ver)), 'y3_comm_info')
            info['communicator_ranks'] = int(n.value)
            info['rccl_version'] = int(ver.value)
        else:
            info['communicator_ranks'] = dist.get_world_size(self.group)      # (the launcher's number; `ranks_summed` below is the communicator's own word)
            if self.backend == 'nccl':
                try:
                    info['rccl_version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
                except Exception as e:          # noqa: BLE001  (diagnostic field only)
                    info['rccl_version'] = 'unavailable: %s' % e
        if self.active and self.model is not None:
            info['ranks_summed'] = self.ranks_summed()
        return info

    def ranks_summed(self):
        """How many ranks the GRADIENT path really sums over: a one-element tensor of ones goes through exactly what a bucket goes
        through (same transport, same stream protocol, same waits) and comes back as the number of contributors -- on every
        transport, including the default one where the process group hides the communicator."""
        dev = self.model.grads.device
        ones = torch.ones(1, dtype=torch.float32, device=dev)
        if self.comm is None:
            dist.all_reduce(ones, op=dist.ReduceOp.SUM, group=self.group)
            return int(round(float(ones.item())))
        own = self._native is not None or self.backend != 'nccl' or os.environ.get('Y3_DP_OWN_STREAM') == '1'
        main = torch.cuda.current_stream(dev)
        if not own:
            w = dist.all_reduce(ones, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            w.wait()
        else:
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(ready)
                w = self._allreduce(ones)
                done = torch.cuda.Event()
                done.record(self.comm)
            if w is not None:
                w.wait()
            main.wait_event(done)
        main.synchronize()
        return int(round(float(ones.item())))

    def broadcast_parameters(self, *tensors):
        """Replicated variables start identical (MirroredStrategy semantics): rank 0's values win."""
        for t in tensors:
            dist.broadcast(t, src=0, group=self.group)

    # -- hooks called by YoloV3.train_step ---------------------------------------
    def begin_step(self):
        self._works = []
        self._last = None
        self._timing = []
        self._wait_span = None
        self._host_wait_s = 0.0

    def _allreduce(self, t):
        """One bucket, issued with the comm stream current.  Returns a Work that still needs a HOST wait, or None."""
        if self._native is not None:
            from ._hip import lib, check
            check(lib.y3_allreduce_sum_f32(self._native, t.data_ptr(), t.numel(), self.comm.cuda_stream), 'y3_allreduce_sum_f32')
            return None
        w = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if self.backend == 'nccl':
            w.wait()          # the comm stream (current) waits for RCCL's stream; the host does not
            return None
        return w              # gloo: finish_step waits on the host

    def on_layer_done(self, layer_idx):
        b = self._by_layer.get(layer_idx)
        if b is None or not self.active:
            return
        lo, hi, _ = b
        g = self.model.grads[lo:hi]
        if self.comm is None:            # host tensors (CPU tests)
            self._works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        own = self._native is not None or self.collect_stats or self.backend != 'nccl' or os.environ.get('Y3_DP_OWN_STREAM') == '1'
        if not own:                      # torch + RCCL: the process group's stream is the comm stream
            self._works.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        main = torch.cuda.current_stream(g.device)
        ready = torch.cuda.Event()
        ready.record(main)               # every gradient of the bucket is enqueued on (or joined into) the compute stream
        with torch.cuda.stream(self.comm):
            self.comm.wait_event(ready)
            if self.collect_stats:
                t0 = torch.cuda.Event(enable_timing=True)
                t0.record(self.comm)
            w = self._allreduce(g)
            if w is not None:
                self._works.append(w)
            done = torch.cuda.Event(enable_timing=self.collect_stats)
            done.record(self.comm)
            if self.collect_stats:
                self._timing.append((t0, done))
        self._last = done

    def finish_step(self):
        if not self.active:
            return
        main = torch.cuda.current_stream(self.model.grads.device) if self.comm is not None else None
        timed = (self.collect_stats or self.time_wait) and main is not None
        if timed:
            before = torch.cuda.Event(enable_timing=True)
            before.record(main)
        t0 = time.perf_counter()
        for w in self._works:
            w.wait()          # nccl: the compute stream waits for RCCL's stream; gloo: host wait (the result is in place when it returns)
        self._host_wait_s = time.perf_counter() - t0
        self._works = []
        if main is not None and self._last is not None:
            main.wait_event(self._last)
        if timed:
            after = torch.cuda.Event(enable_timing=True)
            after.record(main)
            self._wait_span = (before, after)

    def step_stats(self):
        """After a step (and a device synchronize).  With ``collect_stats``: durations of the collectives on this object's OWN comm
        stream (that mode switches the torch + nccl transport from the process group's stream to the own-stream protocol, so the
        figures are labelled ``stats_protocol: own_stream``) and the time the compute stream sat in finish_step's wait.  With
        ``time_wait`` only: that wait alone, in the stream topology the timed steps use (``stats_protocol: timed``).
        gloo waits on the HOST (its events bracket the enqueue only): there ``host_wait_ms`` is the figure and no event spans
        are reported."""
        if not self._timing and self._wait_span is None:
            return None
        torch.cuda.synchronize()
        host_side = self.backend != 'nccl' and self._native is None
        out = dict(stats_protocol='own_stream' if self.collect_stats else 'timed', host_wait_ms=self._host_wait_s * 1e3)
        if self._timing and not host_side:
            durs = [a.elapsed_time(b) for a, b in self._timing]
            out.update(allreduce_ms_sum=float(sum(durs)), allreduce_ms_per_bucket=[round(d, 4) for d in durs])
        if self._wait_span is not None:
            out['allreduce_ms_exposed'] = float(self._wait_span[0].elapsed_time(self._wait_span[1]))
        self.last_stats = out
        return out

    def reduce_sum(self, value):
        """strategy.reduce(SUM, per_replica_loss) (model.py:513)."""
        t = value.detach().clone() if torch.is_tensor(value) else torch.tensor(float(value))
        if self.active:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t

    def mean_moving_stats(self, moving):
        """BN moving statistics are per-replica sync-on-read variables: reading them (a checkpoint save) aggregates by MEAN
        (App. C4) and leaves every replica's own running value untouched.  Returns the mean as a NEW tensor."""
        out = moving.detach().clone()
        if self.active:
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
            out /= self.world_size
        return out
