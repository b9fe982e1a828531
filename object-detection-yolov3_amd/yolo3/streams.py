"""The two extra HIP streams of a training step, bound to hardware queues FIRST.

The ROCm runtime multiplexes HIP streams onto a few hardware queues per priority (GPU_MAX_HW_QUEUES, default 4) and binds
a stream to a queue when the stream is first used; a late stream shares a queue with whatever is least referenced.  The
step relies on two streams really running side by side with the compute stream -- the kernel gradients (model.py, DESIGN
3.1a) and the gradient all-reduce (parallel.py) -- and RCCL / the process group create streams of their own when a
communicator is initialised.  Measured on MI355X (tools/ab_transports.sh, one-rank communicator, collectives forced):
with the process group initialised first the kernel-gradient stream ended up behind the compute stream's queue and the
step went from 18.3 to 21.6 ms (the overlap of the two streams was gone); raising GPU_MAX_HW_QUEUES to 8 repaired that
configuration and wrecked another (30 ms).  So the streams are created and used here, once per device, before any
communicator exists: call ``reserve()`` right after ``torch.cuda.set_device`` and before ``init_process_group``.
"""
import torch

_reserved = {}


def reserve(device=None):
    """(side stream for the kernel gradients, comm stream for the collectives) of ``device``; created and given one
    command each on first call."""
    dev = torch.device('cuda', torch.cuda.current_device()) if device is None else torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _reserved:
        torch.zeros(1, device=dev)             # the compute (default) stream binds first
        side, comm = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        for s in (side, comm):
            with torch.cuda.stream(s):
                torch.zeros(8, device=dev).add_(1.0)
        torch.cuda.synchronize(dev)
        _reserved[key] = (side, comm)
    return _reserved[key]
