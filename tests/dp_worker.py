"""Rank process of tests/test_gpu_dist.py: one data-parallel training step of the real model (gloo backend, so two ranks
can share cuda:0; the collective is elementwise either way) and a dump of what the step left behind.
usage: dp_worker.py OUT_DIR IMG N_PER_RANK SEED [BACKEND [TRANSPORT]]   (RANK / WORLD_SIZE / MASTER_* in the environment)

BACKEND nccl (= RCCL) with WORLD_SIZE 1 is the single-GPU rehearsal of the real transport: DataParallel then runs with
force_collective, so every bucket goes through the process group's stream / event machinery although there is one rank.
Three steps are taken in that mode (a race between a collective and the next step's kernels needs a next step)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'object-detection-yolov3_amd'), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np                      # noqa: E402
import torch                            # noqa: E402
import torch.distributed as dist       # noqa: E402


def make_case(img, n_total, seed):
    """The full global batch (every rank builds the same one and takes its slice)."""
    from oracle import model as om     # checker-side weights: identical initial values for the product and the oracle
    from test_gpu_kernels import _labels
    anchors, K = [(64, 384), (384, 64)], 2
    params = om.init_params(3, len(anchors), K, seed=seed)
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(n_total, 3, img, img, generator=g)
    gts = _labels(np.random.default_rng(seed), n_total, img, anchors, K, per_image=3)
    return anchors, K, params, images, gts


def main():
    out_dir, img, n, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    backend = sys.argv[5] if len(sys.argv) > 5 else 'gloo'
    transport = sys.argv[6] if len(sys.argv) > 6 else 'torch'
    steps = 3 if world == 1 else 1
    torch.cuda.set_device(0)
    from yolo3 import streams
    streams.reserve()
    if backend == 'nccl':
        dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
    else:
        dist.init_process_group(backend)
    from yolo3.model import YoloV3
    from yolo3.parallel import DataParallel
    anchors, K, params, images, gts = make_case(img, n * world, seed)
    yolo = YoloV3(n * world, [img, img, 3], K, anchors, learning_rate=1e-3)
    if rank == 0:
        yolo.set_weights(params)        # the other ranks keep their own random init until the broadcast
    strategy = DataParallel(bucket_mb=8.0, force_collective=(world == 1), transport=transport, collect_stats=(world == 1))
    strategy.attach(yolo)
    strategy.broadcast_parameters(yolo.params, yolo.moving)
    yolo._refresh_transposed()
    sl = slice(rank * n, (rank + 1) * n)
    for _ in range(steps):
        loss = yolo.dist_train_step(strategy, (images[sl].cuda(), [torch.from_numpy(x[sl]).cuda() for x in gts]))
    torch.cuda.synchronize()
    stats = strategy.step_stats() or {}
    info = strategy.comm_info()
    np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), grads=yolo.grads.cpu().numpy(), params=yolo.params.cpu().numpy(),
             moving=yolo.moving.cpu().numpy(), adam_m=yolo.adam_m.cpu().numpy(), loss=float(loss), buckets=len(strategy.buckets),
             collectives=len(stats.get('allreduce_ms_per_bucket', [])), communicator_ranks=int(info['communicator_ranks']),
             rccl_version=str(info.get('rccl_version')), ranks_summed=int(info['ranks_summed']))
    strategy.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
