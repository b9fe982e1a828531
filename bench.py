#!/usr/bin/env python3
"""bench.py -- images/sec of the YOLOv3 training step on MI355X.

Workload (BASELINE.json metric): one ``train.py`` step = forward (training-mode
BatchNorm) + loss + backward + Keras Adam (+ RCCL gradient all-reduce when
N > 1), per-GPU batch 8, 416x416x3, anchors [(64,384),(384,64)], 2 classes,
fp32, synthetic N(0,1) images / random labels / Glorot weights, all resident
in HBM before the timed region.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          # spawns its own N ranks (fresh processes, before any GPU call)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  ``roofline`` covers the dominant kernels (the
MFMA implicit-GEMM conv family: forward, dgrad, wgrad): algorithmic conv FLOPs
of one step / summed kernel time of those launches, timed with HIP events on
the launch stream in an extra instrumented step.  ``cpu_baseline`` times the
CPU oracle (torch-CPU restatement of the reference graph; TensorFlow is not
available) on a bounded sample, rank 0 and N = 1 only.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'object-detection-yolov3_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np   # noqa: E402
import torch         # noqa: E402

ANCHORS = [(64, 384), (384, 64)]
K = 2
IMG = 416
BATCH = 8
FP32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0    # same guide: dense bf16 MFMA (not the 2:1 sparsity figure)


def tiled_4k(use_graph, precisions=('bf16', 'fp32')):
    """BASELINE.json configs[4]: inference_tiled on one synthetic 4096 x 4096 x 3 uint8 image (SURVEY 8d seed 4), 608 x 608
    tiles (100 of them, 96-px ghost border), random-init weights: upload -> GPU tiling + per-tile z-score -> network in
    batches of 25 -> decode -> GPU NMS -> host merge.  Times the whole function per image, fp32 and bf16 conv paths."""
    import contextlib
    import io
    import inference_tiled
    from yolo3.model import YoloV3
    y = YoloV3(25, [608, 608, 3], K, ANCHORS, seed=1, use_graph=use_graph)
    big = np.random.default_rng(4).integers(0, 256, (4096, 4096, 3), dtype=np.uint8)
    tile_fl = conv_flops_per_image(y.specs, 608)[0]
    tb = int(os.environ.get('Y3_TILED_BATCH', '0')) or None      # experiments: fixed tiles per launch instead of the planned batches
    out = {'image': [4096, 4096, 3], 'tile': [608, 608], 'tiles': 100,
           'batches': {'bf16': inference_tiled.plan_tile_batches(100, [608, 608]), 'fp32': [25, 25, 25, 25]}}
    for prec in precisions:      # bf16 first: its buffers then come out of untouched GPU memory (measured: the figure is bimodal, 30 / 38 ms, with the memory the allocator hands out)
        y.inference_precision = prec
        mdl = y.get_keras_model()
        with contextlib.redirect_stdout(io.StringIO()):
            inference_tiled.inference_image_tiled(mdl, big, [608, 608], 32, batch_size=tb)
            torch.cuda.synchronize()
            times = []
            for _ in range(5):      # every image timed on its own: the path is host-driven (upload, launches on two streams, merge),
                t1 = time.perf_counter()      # and one slow repetition on a shared box moved a 3-image mean from 30.4 to 38.8 ms
                inference_tiled.inference_image_tiled(mdl, big, [608, 608], 32, batch_size=tb)
                torch.cuda.synchronize()
                times.append(time.perf_counter() - t1)
        t = sorted(times)[len(times) // 2]
        out[prec] = {'ms_per_image_end_to_end': t * 1e3, 'statistic': 'median of 5 images', 'ms_min': min(times) * 1e3, 'ms_max': max(times) * 1e3,
                     'tiles_per_s': 100 / t, 'conv_tflops_end_to_end': tile_fl * 100 / t / 1e12}
    del y
    torch.cuda.empty_cache()
    return out


def conv_macs_per_image(specs, img):
    """MACs of every conv layer for one image, creation order."""
    # replay the strides: spatial size of each layer's OUTPUT (creation order)
    seq = [1, 2, 2, 2, 4] + [4] * 4 + [8] + [8] * 16 + [16] + [16] * 16 + [32] + [32] * 8
    seq += [32] * 7 + [32] + [16] * 7 + [16] + [8] * 7
    assert len(seq) == len(specs)
    return [(img // st) ** 2 * sp.k * sp.k * sp.cin * sp.cout for sp, st in zip(specs, seq)]


def conv_flops_per_image(specs, img):
    """2*MAC of every conv layer: (fwd, train = fwd + dgrad + wgrad, no dgrad for conv1)."""
    macs = conv_macs_per_image(specs, img)
    fwd = 2 * sum(macs)
    return fwd, 3 * fwd - 2 * macs[0]


def measure_train(yolo, strategy, inputs, steps, warmup, barrier):
    """W untimed + K timed steps between barriers: seconds for the K steps, last loss."""
    loss = None
    for _ in range(warmup):
        loss = yolo.dist_train_step(strategy, inputs)
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = yolo.dist_train_step(strategy, inputs)
    barrier()
    return time.perf_counter() - t0, loss


def conv_family_roofline(yolo, plan, busy_s):
    """Work of one step's conv launches by the matrix instruction they issue, and the fraction of the matrix pipe's peak
    the family reaches over its busy time.  Launches with Y3_CONV_X3 (conv_x3.hip) issue SIX v_mfma_f32_32x32x16_bf16 products per
    fp32 product: their work counts as 6 x algorithmic FLOPs against the dense bf16 peak (2 500 TFLOP/s) -- the peak of the
    instruction actually issued (VERDICT r3 ruling); the others issue v_mfma_f32_32x32x2_f32 (157.3 TFLOP/s).
    frac = (time the launches would take at those peaks) / (busy time of the family)."""
    macs = conv_macs_per_image(yolo.specs, IMG)
    total = (3 * 2 * sum(macs) - 2 * macs[0]) * BATCH
    x3 = 2 * BATCH * (sum(macs[i] for i in plan.x3_fwd) + sum(macs[i] for i in plan.x3_dgrad) + sum(macs[i] for i in getattr(plan, 'x3_wgrad', [])))
    f32 = total - x3
    t_peak = (6.0 * x3 / (BF16_MFMA_PEAK_TFLOPS * 1e12)) + f32 / (FP32_MFMA_PEAK_TFLOPS * 1e12)
    return {'flops_per_step': total, 'flops_x3_path': x3, 'flops_f32_mfma_path': f32, 'frac': t_peak / busy_s,
            'achieved_bf16_equivalent': (6.0 * x3 + f32 * BF16_MFMA_PEAK_TFLOPS / FP32_MFMA_PEAK_TFLOPS) / busy_s / 1e12,
            'algorithmic_tflops': total / busy_s / 1e12,
            'launches_x3': {'fwd': len(plan.x3_fwd), 'dgrad': len(plan.x3_dgrad), 'wgrad': len(getattr(plan, 'x3_wgrad', []))}}


def synth_labels(rng, n):
    from yolo3.imagereader import format_boxes
    labs = [[], [], []]
    for _ in range(n):
        k = int(rng.integers(1, 5))
        wh = rng.integers(40, 301, (k, 2))
        xy = np.stack([rng.integers(0, IMG - wh[:, 0] + 1), rng.integers(0, IMG - wh[:, 1] + 1)], 1)
        boxes = np.concatenate([xy, wh, rng.integers(0, K, (k, 1))], 1).astype(np.int32)
        lab = format_boxes(boxes, (IMG, IMG, 3), ANCHORS, K)
        for i in range(3):
            labs[i].append(lab[i])
    return [np.stack(l) for l in labs]


def timed_conv_pass(yolo, plan):
    """One eager step, launched exactly as train_step launches it (kernel gradients on the plan's second stream when it
    has one), with a HIP-event pair around every MFMA conv entry on the stream that entry runs on.  Returns
    (busy seconds, per-entry dict): busy = length of the UNION of the conv intervals -- with two streams the kernel
    gradient of a layer overlaps its data gradient, so summing durations would count that wall time twice."""
    from yolo3._hip import check
    conv_fns = {'y3_conv2d_fwd': 'conv2d_fwd', 'y3_conv2d_dgrad': 'conv2d_dgrad', 'y3_conv2d_dgrad_bn': 'conv2d_dgrad',
                'y3_conv2d_wgrad': 'conv2d_wgrad', 'y3_conv2d_wgrad_x': 'conv2d_wgrad'}
    main = torch.cuda.current_stream()
    st = main.cuda_stream
    base = torch.cuda.Event(enable_timing=True)
    base.record(main)
    pairs = []

    def timed(fn, args, stream_obj):
        name = conv_fns.get(getattr(fn, '__name__', ''))
        if not name:
            check(fn(*args, stream_obj.cuda_stream), 'launch')
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream_obj)
        check(fn(*args, stream_obj.cuda_stream), name)
        b.record(stream_obj)
        pairs.append((name, a, b))

    for lst in (plan.fwd, None, plan.bwd):
        if lst is None:
            plan.run_loss(st)
            continue
        for fn, args in lst:
            if fn == 'layer_done':
                continue
            if fn == 'record':
                plan.events[args].record(main)
            elif fn == 'main_wait':
                main.wait_event(plan.events[args])
            elif fn == 'side_call':
                f2, a2, e_wait, e_done = args
                plan.side.wait_event(plan.events[e_wait])
                timed(f2, a2, plan.side)
                plan.events[e_done].record(plan.side)
            else:
                timed(fn, args, main)
    torch.cuda.synchronize()
    per, spans = {}, []
    for name, a, b in pairs:
        d = per.setdefault(name, [0.0, 0])
        d[0] += a.elapsed_time(b) * 1e-3
        d[1] += 1
        spans.append((base.elapsed_time(a) * 1e-3, base.elapsed_time(b) * 1e-3))
    spans.sort()
    busy, cur_s, cur_e = 0.0, None, None
    for s0, e0 in spans:
        if cur_e is None or s0 > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s0, e0
        else:
            cur_e = max(cur_e, e0)
    if cur_e is not None:
        busy += cur_e - cur_s
    return busy, per


def cpu_baseline(seed, budget_s=25.0):
    """The oracle's train step (torch-CPU fp32) on the same shapes; bounded sample."""
    from oracle import model as om
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 16)))
    cores = torch.get_num_threads()
    params = om.init_params(3, len(ANCHORS), K, seed=seed)
    rng = np.random.default_rng(seed)
    g = torch.Generator().manual_seed(seed)
    # probe with a small batch to size the sample
    def run(bs, steps):
        net = om.Net(params, 3, len(ANCHORS), K, dtype=torch.float32, requires_grad=True)
        adam = om.AdamState(net.trainable(), 1e-4)
        images = torch.randn(bs, 3, IMG, IMG, generator=g)
        gts = [torch.from_numpy(x) for x in synth_labels(rng, bs)]
        t0 = time.time()
        for _ in range(steps):
            om.train_step(net, adam, images, gts, (IMG, IMG, 3), ANCHORS, K, bs)
        return time.time() - t0
    t1 = run(1, 1)                    # includes first-touch overheads; only used to size the sample
    bs = BATCH if t1 * BATCH <= budget_s else max(1, int(budget_s / max(t1, 1e-3)))
    t = run(bs, 1)                    # warm: allocator and thread pool settled
    steps = max(1, min(8, int(0.6 * budget_s / max(t, 1e-3))))          # about 10-20 s of CPU work in the timed sample
    t = run(bs, steps)
    return dict(value=bs * steps / t, unit='images/s', cores=cores, kind='port',
                sample='oracle/model.py train_step (torch-CPU fp32 restatement of model.py:481-508; TensorFlow unavailable), '
                       '%d step(s) at batch %d, 416x416, %.1f s' % (steps, bs, t))


def spawn_ranks(n, timeout_s=1500.0):
    """`python bench.py --gpus N` outside a launcher: start N fresh rank processes of this script (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* in their environment, exactly what torch.distributed.run would set) and relay rank 0's JSON line.
    This process has not touched the GPU (importing torch does not initialise HIP) and never does: it only waits.  Every
    child writes to a file of its own (no pipe can fill up); all children are polled, and when one exits non-zero -- or the
    time limit passes -- the others are terminated instead of sitting in the rendezvous timeout."""
    import socket
    import subprocess
    import tempfile
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    tmp = tempfile.mkdtemp(prefix='y3bench_')
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        out = open(os.path.join(tmp, 'rank%d.out' % r), 'w')
        err = open(os.path.join(tmp, 'rank%d.err' % r), 'w')
        logs.append((out, err))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out, stderr=err))
    deadline = time.time() + timeout_s
    failed = None
    while True:
        codes = [p.poll() for p in procs]
        bad = [i for i, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = 'rank %d exited with code %d' % (bad[0], codes[bad[0]])
        elif time.time() > deadline:
            failed = 'time limit of %.0f s passed' % timeout_s
        if failed or all(c == 0 for c in codes):
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    for out, err in logs:
        out.close()
        err.close()
    sys.stdout.write(open(os.path.join(tmp, 'rank0.out')).read())
    sys.stdout.flush()
    for r in range(n):      # stderr of every rank is relayed (RCCL warnings of rank > 0 would otherwise be lost): a short tail on success
        tail = open(os.path.join(tmp, 'rank%d.err' % r)).read()[-(1500 if failed else 400):]
        if tail.strip():
            sys.stderr.write('--- rank %d stderr (tail) ---\n%s\n' % (r, tail))
    if failed:
        raise SystemExit('bench ranks failed: %s (exit codes %s); per-rank logs kept in %s' % (failed, [p.returncode for p in procs], tmp))
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--graph', action='store_true', help='replay the training step as one HIP graph (single stream) instead of host launches with the kernel gradients on a second stream')
    ap.add_argument('--no-graph', action='store_true', help='(default now; kept for older command lines)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-f32-reference', action='store_true', help='skip the second training measurement on the plain fp32-MFMA path')
    ap.add_argument('--no-tiled', action='store_true', help='skip the tiled 4k x 4k inference measurement (BASELINE.json configs[4])')
    ap.add_argument('--tiled-only', action='store_true', help='(internal) run only the tiled 4k x 4k measurement and print its JSON: the full bench runs it in a process of its own')
    ap.add_argument('--no-inference', action='store_true', help='skip the secondary inference measurement (config: bs=8 fp32 predict + NMS)')
    ap.add_argument('--bucket-mb', type=float, default=32.0)
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend: nccl (= RCCL, one GPU per rank) or gloo (rehearsal: ranks may share a GPU)")
    ap.add_argument('--check-replicas', action='store_true', help='after the run, verify that every rank holds identical weights')
    ap.add_argument('--transport', default='torch', choices=['torch', 'native'], help="gradient all-reduce through torch.distributed (default) or RCCL called directly through the C ABI (y3_comm_*)")
    ap.add_argument('--force-collective', action='store_true', help='(rehearsal) with one rank: initialise the process group anyway and all-reduce every bucket')
    args = ap.parse_args()
    if args.tiled_only:
        torch.cuda.set_device(0)
        print(json.dumps(tiled_4k(True)), flush=True)
        return 0

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus > 1 and 'RANK' not in os.environ:
        return spawn_ranks(args.gpus)              # plain `python bench.py --gpus N`: be our own launcher
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit('--gpus %d but the launcher started WORLD_SIZE=%d ranks' % (args.gpus, world))
    ndev = torch.cuda.device_count()               # counting devices does not initialise the GPU
    isolated = any(os.environ.get(k) for k in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'))
    if world > 1 and args.backend == 'nccl' and ndev < int(os.environ.get('LOCAL_WORLD_SIZE', str(world))) and not (isolated and ndev == 1):
        # (a launcher that hands every rank its OWN device through *_VISIBLE_DEVICES shows exactly one device per rank: that is fine;
        # a mask that shows several but fewer than the ranks would put two ranks on one device)
        raise SystemExit('backend nccl (= RCCL) needs one GPU per rank: %d visible, %d ranks (use --backend gloo to rehearse on fewer GPUs)' % (ndev, world))
    torch.cuda.set_device(local_rank % max(1, ndev))
    import torch.distributed as dist
    from yolo3.model import YoloV3
    from yolo3 import streams
    streams.reserve()          # the step's side / comm streams take their hardware queues before RCCL creates its own (yolo3/streams.py)
    strategy = None
    if world > 1 or args.force_collective:
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if world == 1:
            import socket
            s_ = socket.socket()
            s_.bind(('127.0.0.1', 0))
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(s_.getsockname()[1]))
            s_.close()
            os.environ.setdefault('RANK', '0')
            os.environ.setdefault('WORLD_SIZE', '1')
        import datetime
        pg_timeout = datetime.timedelta(seconds=300)      # a collective that hangs aborts the rank with a message instead of sitting in the driver's limit
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank % max(1, ndev)), timeout=pg_timeout)
        else:
            dist.init_process_group(args.backend, timeout=pg_timeout)
        from yolo3.parallel import DataParallel
        strategy = DataParallel(bucket_mb=args.bucket_mb, force_collective=args.force_collective, transport=args.transport)

    global_batch = BATCH * world
    use_graph = args.graph and not args.no_graph and world == 1
    yolo = YoloV3(global_batch, [IMG, IMG, 3], K, ANCHORS, learning_rate=1e-4, seed=1, use_graph=use_graph)
    if strategy is not None:
        strategy.attach(yolo)
        strategy.broadcast_parameters(yolo.params, yolo.moving)
        yolo._refresh_transposed()
    g = torch.Generator().manual_seed(100 + rank)
    images = torch.randn(BATCH, 3, IMG, IMG, generator=g).cuda()
    gts = [torch.from_numpy(x).cuda() for x in synth_labels(np.random.default_rng(3 + rank), BATCH)]
    inputs = (images, gts)

    def barrier():
        torch.cuda.synchronize()
        if strategy is not None:
            dist.barrier()
        torch.cuda.synchronize()

    dt, loss = measure_train(yolo, strategy, inputs, args.steps, args.warmup, barrier)
    # host cost of one step: queue drained first, so that the figure is launch work and not back-pressure from a full queue
    # (outside the timed region; tools/host_profile.py: ~3.1 ms, i.e. the host could feed a step six times as fast)
    t1 = time.perf_counter()
    loss_extra = yolo.dist_train_step(strategy, inputs)
    t_issued = time.perf_counter() - t1
    barrier()
    del loss_extra
    comm = None
    if strategy is not None:
        # two instrumented steps (outside the timed region).  (1) the TIMED stream topology, untouched: only a pair of events on the
        # compute stream around finish_step's wait = the exposed, un-overlapped part of the collectives as the timed steps see it.
        # (2) the own-stream protocol (what collect_stats switches the torch + nccl transport to): events around every bucket's
        # all-reduce on the comm stream too -- how long the collectives themselves ran.  Both labelled with their protocol.
        strategy.time_wait = True
        yolo.dist_train_step(strategy, inputs)
        barrier()
        st_timed = strategy.step_stats() or {}
        strategy.time_wait = False
        strategy.collect_stats = True
        yolo.dist_train_step(strategy, inputs)
        barrier()
        st_ = strategy.step_stats() or {}
        strategy.collect_stats = False
        comm = strategy.comm_info()      # backend, transport, world size, ranks_summed (ones through the gradient path), RCCL version
        comm.update(st_)
        comm['timed_topology'] = st_timed
        comm['payload_mb_per_step'] = sum(hi - lo for lo, hi, _ in strategy.buckets) * 4 / 1e6
        # host side per rank: launch work of one step (max over ranks below), threads torch may use, CPUs of the box
        comm['host'] = {'torch_threads_per_rank': torch.get_num_threads(), 'cpus': os.cpu_count(), 'local_world_size': int(os.environ.get('LOCAL_WORLD_SIZE', str(world))),
                        'reader_processes_per_rank': 0, 'note': 'bench.py feeds resident synthetic batches: no reader processes; train.py starts min(--reader_count, cpus // LOCAL_WORLD_SIZE - 1) per rank'}
    if world > 1:
        tt = torch.tensor([dt, t_issued], dtype=torch.float64, device='cuda')
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, t_issued = float(tt[0].item()), float(tt[1].item())      # both: max over ranks
        if args.check_replicas:
            ref = yolo.params.clone()
            dist.broadcast(ref, src=0)
            same = torch.tensor([1.0 if torch.equal(ref, yolo.params) else 0.0], device='cuda')
            dist.all_reduce(same, op=dist.ReduceOp.MIN)
            if float(same.item()) != 1.0:
                raise SystemExit('replicas diverged: weights differ between ranks')
    loss_val = float(loss) if loss is not None else float('nan')
    if not np.isfinite(loss_val):
        raise SystemExit('loss is not finite: %r' % loss_val)

    # instrumented pass: HIP events around every conv launch (after the timed region)
    fwd_fl, train_fl = conv_flops_per_image(yolo.specs, IMG)
    plan = yolo._plan(BATCH, True)
    conv_s, per = timed_conv_pass(yolo, plan)
    conv_s2, per = timed_conv_pass(yolo, plan)        # second pass: caches / clocks settled
    conv_s = min(conv_s, conv_s2)
    roof = conv_family_roofline(yolo, plan, conv_s)
    # The same step on the plain path (every conv on v_mfma_f32_32x32x2_f32), measured in the same run: required beside a line whose
    # arithmetic is the bf16-piece form (one GPU only: the reference is about the kernels, not about the collective)
    f32_ref = None
    if world == 1 and yolo.conv_arithmetic != 'f32' and not args.no_f32_reference:
        yref = YoloV3(global_batch, [IMG, IMG, 3], K, ANCHORS, learning_rate=1e-4, seed=1, use_graph=use_graph, conv_arithmetic='f32')
        dt_ref, _ = measure_train(yref, None, inputs, args.steps, args.warmup, barrier)
        pref = yref._plan(BATCH, True)
        timed_conv_pass(yref, pref)
        cs_ref, _ = timed_conv_pass(yref, pref)
        f32_ref = {'value': global_batch * args.steps / dt_ref, 'ms_per_step': dt_ref / args.steps * 1e3,
                   'frac': train_fl * BATCH / cs_ref / 1e12 / FP32_MFMA_PEAK_TFLOPS, 'kernel_ms_per_step': cs_ref * 1e3,
                   'note': 'same model, conv_arithmetic=f32: every conv launch on v_mfma_f32_32x32x2_f32; frac = algorithmic flops / family busy time / 157.3'}
        del yref, pref
        torch.cuda.empty_cache()
    # HBM-side traffic of the same kernel family comes from separate rocprofv3 --pmc passes (FETCH_SIZE x2 on gfx950,
    # WRITE_SIZE; see profiles/README.md): bench.py cannot run the profiler on itself, so it reports the figure committed in
    # the SAME round as the kernels it times (profiles/rNN_traffic.json, newest first) and says which file it was
    traffic, traffic_src = None, None
    for tag in ('r04', 'r03', 'r02', 'r01'):
        try:
            with open(os.path.join(ROOT, 'profiles', '%s_traffic.json' % tag)) as fh:
                traffic = json.load(fh).get('conv_family_hbm_bytes_per_step')
                traffic_src = 'profiles/%s_traffic.json' % tag
                break
        except (OSError, ValueError):
            pass

    # secondary (BASELINE.json configs[1]): inference bs=8 416x416 fp32 -- z-scored batch -> conv stacks with folded BN ->
    # decode -> clip + small-box filter + class-wise NMS, all on the GPU; reported beside the headline metric
    infer = infer16 = tiled = None
    if world == 1 and not args.no_inference:
        from yolo3 import bbox_utils
        # inference replays forward + decode as ONE HIP graph (the launch list is static): ~80 launches of 20-130 us each are
        # launch-gap bound when issued one by one.  Same architecture / shapes, its own random weights.
        ymodel = yolo
        yolo = YoloV3(BATCH, [IMG, IMG, 3], K, ANCHORS, seed=1, use_graph=True)
        for _ in range(3):
            rows = yolo.predict(images)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_inf = 20
        for _ in range(n_inf):
            rows = yolo.predict(images)
        torch.cuda.synchronize()
        t_fwd = (time.perf_counter() - t1) / n_inf
        bbox_utils.nms_device(rows, 32.0, clip_wh=(IMG, IMG))
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_inf):
            bbox_utils.nms_device(rows, 32.0, clip_wh=(IMG, IMG))
        torch.cuda.synchronize()
        t_nms = (time.perf_counter() - t1) / n_inf
        # the HBM-bound tail of the path on its own (north_star: "achieved HBM GB/s for the decode/NMS path"): HIP events on
        # the launch stream around y3_decode_fwd and y3_nms_per_class, 50 launches each; algorithmic bytes per SURVEY 8d
        # (decode reads and writes N*Nb*(5+K)*4 B, NMS reads the same rows); rocprofv3 figures: profiles/r03_hbm_path.md
        iplan = yolo._plan(BATCH, False)
        cur = torch.cuda.current_stream()

        def ev_us(fn, reps=50):
            fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(cur)
            for _ in range(reps):
                fn()
            e1.record(cur)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / reps
        decode_us = ev_us(lambda: iplan.run_decode(cur.cuda_stream))
        nms_us = ev_us(lambda: bbox_utils.nms_device(rows, 32.0, clip_wh=(IMG, IMG)))
        row_bytes = BATCH * rows.shape[1] * rows.shape[2] * 4
        infer = {'images_per_s_forward_decode': BATCH / t_fwd, 'ms_forward_decode': t_fwd * 1e3, 'ms_nms_batch8': t_nms * 1e3, 'launch': 'hip-graph',
                 'decode_us': decode_us, 'decode_gbps': 2 * row_bytes / decode_us / 1e3, 'nms_us': nms_us, 'nms_gbps': row_bytes / nms_us / 1e3,
                 'decode_nms_bytes': {'rows_bytes': row_bytes, 'decode': '2 x rows (read feature maps, write rows)', 'nms': '1 x rows + keep lists',
                                      'note': '1.6 MB per launch: launch / latency bound, far below the 8 TB/s HBM peak by construction; us is the figure to compare'},
                 'forward_tflops': fwd_fl * BATCH / t_fwd / 1e12, 'conv_arithmetic': yolo.conv_arithmetic, 'x3_forward_launches': len(iplan.x3_fwd),
                 'forward_frac_of_fp32_mfma_peak': fwd_fl * BATCH / t_fwd / 1e12 / FP32_MFMA_PEAK_TFLOPS,
                 'frac_note': 'algorithmic forward FLOPs over the whole forward + decode time against 157.3 TFLOP/s; with conv_arithmetic x3 the named launches '
                              'run on the bf16 matrix pipe (6 products per fp32 product), so this ratio is a speed statement, not a utilisation of that peak'}
        # the same batch on the bf16 conv path (v_mfma_f32_32x32x16_bf16, fp32 accumulate / heads / decode / NMS)
        for _ in range(3):
            yolo.predict(images, precision='bf16')
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_inf):
            yolo.predict(images, precision='bf16')
        torch.cuda.synchronize()
        t16 = (time.perf_counter() - t1) / n_inf
        infer16 = {'images_per_s_forward_decode': BATCH / t16, 'ms_forward_decode': t16 * 1e3, 'forward_tflops': fwd_fl * BATCH / t16 / 1e12,
                   'forward_frac_of_bf16_mfma_peak': fwd_fl * BATCH / t16 / 1e12 / BF16_MFMA_PEAK_TFLOPS}
        del yolo
        yolo = ymodel
        if not args.no_tiled:
            # In a process of its own: the path streams 25 x 608^2 activations, and buffers carved out of GPU memory that this
            # process has already churned (training plan, two inference plans, freed and re-used segments) ran it 25 % slower
            # (37.5 vs 29.8 ms per image, either launch mode) than the same code in a fresh process -- which is how
            # inference_tiled.py is used.  A child process, not an exec: this process has initialised the GPU.
            import subprocess
            res = subprocess.run([sys.executable, os.path.abspath(__file__), '--tiled-only'], capture_output=True, text=True, timeout=600)
            lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
            if res.returncode != 0 or not lines:
                raise SystemExit('tiled 4k measurement failed (rc %d): %s' % (res.returncode, res.stderr[-400:]))
            tiled = json.loads(lines[-1])
            tiled['process'] = 'own (fresh GPU context)'
            # ... and once in THIS process, after dropping every plan of the runs above and handing the cached segments back to
            # the driver (say what the figure depends on): the same code in memory this process has already churned
            import gc
            ymodel._plans.clear()
            del iplan, rows
            gc.collect()
            torch.cuda.empty_cache()
            tiled['bf16_in_bench_process_after_empty_cache'] = tiled_4k(True, precisions=('bf16',))['bf16']

    if rank == 0:
        out = {
            'metric': 'images/sec (train fwd+bwd) bs=8 416x416',
            'value': global_batch * args.steps / dt,
            'unit': 'images/s',
            'n_gpus': world,
            'steps': args.steps,
            'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3,
            'host_issue_ms_one_step_empty_queue': t_issued * 1e3,
            'higher_is_better': True,
            'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32',
            'arithmetic': ('3xbf16 split operands, 6 partial products, fp32 accumulate (v_mfma_f32_32x32x16_bf16) on %d forward / %d data-gradient / %d kernel-gradient '
                           'launches; v_mfma_f32_32x32x2_f32 on the others' % (roof['launches_x3']['fwd'], roof['launches_x3']['dgrad'], roof['launches_x3']['wgrad']))
                          if roof['flops_x3_path'] else 'v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain)',
            'data': 'synthetic',
            'config': {'workload': 'train.py step (model.py:481-508): fwd + loss + bwd + Keras Adam%s, batch 8 per GPU, 416x416x3, '
                                   'anchors [(64,384),(384,64)], 2 classes, fp32 convs on the matrix cores' % (' + RCCL grad all-reduce' if world > 1 else ''),
                       'global_batch': global_batch, 'per_gpu_batch': BATCH, 'image': [IMG, IMG, 3],
                       'launch': 'hip-graph' if use_graph else 'host launches, kernel gradients on a second stream', 'parallelism': 'dp%d' % world},
            'roofline': {'bound': 'mfma', 'achieved': roof['achieved_bf16_equivalent'], 'peak': BF16_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': roof['frac'], 'traffic': traffic, 'traffic_source': traffic_src,
                         'achieved_note': 'matrix-instruction work over the union of the conv family\'s busy intervals, in bf16-MFMA-equivalent TFLOP/s: launches on the '
                                          'x3 path count 6 x their algorithmic FLOPs (six v_mfma_f32_32x32x16_bf16 products per fp32 product, peak 2 500), launches on '
                                          'v_mfma_f32_32x32x2_f32 count their algorithmic FLOPs x 2500/157.3 (that instruction\'s peak is 157.3); frac = achieved / peak '
                                          '= (time at the peak of the instruction each launch issues) / busy time.  NOT algorithmic FLOPs against 157.3.',
                         'algorithmic_tflops': roof['algorithmic_tflops'], 'flops_x3_path': roof['flops_x3_path'], 'flops_f32_mfma_path': roof['flops_f32_mfma_path'],
                         'kernel': 'conv_x3_kernel + conv_igemm_fast_kernel + conv_wgrad_kernel (MFMA implicit-GEMM conv fwd/dgrad/wgrad; split-K and kernel gradients of <= 8 pixel splits reduced in-kernel, the others followed by slab_reduce_kernel inside the same entry), %d entry calls/step' % sum(v[1] for v in per.values()),
                         'peak_note': 'bf16 peak = 256 CUs x 4 SIMDs x 1024 FLOP/clk x 2.4 GHz (dense); fp32-MFMA peak 157.3 = 64 FLOP/clk; the chip holds 1.9-2.0 GHz under the x3 kernels and 2.2-2.4 GHz under the fp32-MFMA kernels (tools/probe/conv_timing, profiles/r04_clock_ablate.txt)',
                         'flops_per_step': train_fl * BATCH, 'kernel_ms_per_step': conv_s * 1e3,
                         'by_entry_ms': {k: round(v[0] * 1e3, 3) for k, v in per.items()}},
            'step_flop_rate_tflops': train_fl * BATCH / (dt / args.steps) / 1e12,
            'final_loss': loss_val,
        }
        if f32_ref is not None:
            out['fp32_mfma_reference'] = f32_ref
        if infer is not None:
            out['inference_bs8_fp32'] = infer
        if infer16 is not None:
            out['inference_bs8_bf16'] = infer16
        if tiled is not None:
            out['tiled_4k_608'] = tiled
        if comm is not None:
            out['comm'] = comm
        if world > 1:
            out['config']['backend'] = args.backend
            out['replicas_identical'] = True if args.check_replicas else None
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(seed=1)
        print(json.dumps(out), flush=True)
    if strategy is not None:
        strategy.close()
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
