"""The data-parallel step (model.py:492,510-515, train.py:38-41) with real gradients: two ranks, each
with its own half of a global batch, bucketed async all-reduce hooked into backward.  RCCL needs one GPU per rank and
the test box has one, so the ranks use the gloo backend and share cuda:0 -- the collective is an elementwise SUM
either way; the RCCL N > 1 path itself stays unverified until the driver's 8-GPU run."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'object-detection-yolov3_amd')


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_two_ranks(out_dir, img, n, seed, wgrad_stream):
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   Y3_WGRAD_STREAM=wgrad_stream)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'tests', 'dp_worker.py'), out_dir, str(img), str(n), str(seed)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-4000:]
    return [np.load(os.path.join(out_dir, 'rank%d.npz' % r)) for r in range(2)]


def test_two_rank_step_sums_gradients_and_matches_oracle(tmp_path):
    """(1) after the all-reduce every rank holds g0 + g1 BIT FOR BIT, where g_r is what a single process computes on
    rank r's half with the loss divided by the GLOBAL batch (SUM, no averaging: Q8) -- a wrong scale factor or a bucket
    reduced before its kernel gradients were complete cannot pass;  (2) the weights after Adam equal Adam applied to
    that sum, bit for bit, on both ranks;  (3) BatchNorm statistics stay per replica;  (4) the reduced loss is the sum of
    the per-replica losses;  (5) the summed gradient agrees with the CPU oracle evaluated per shard (per-replica BN
    statistics) and summed, in the relative-L2 yardstick of test_gpu_model.py;  (6) kernel gradients on the second
    stream (Y3_WGRAD_STREAM=1) or on the main stream (=0): identical bits."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from dp_worker import make_case
    from oracle import model as om
    from yolo3.model import YoloV3
    img, n, seed = 96, 4, 11
    anchors, K, params, images, gts = make_case(img, 2 * n, seed)
    # single-process references on each half, global batch = 2n
    singles = []
    for r in range(2):
        m = YoloV3(2 * n, [img, img, 3], K, anchors, learning_rate=1e-3)
        m.set_weights(params)
        sl = slice(r * n, (r + 1) * n)
        loss = float(m.train_step((images[sl].cuda(), [torch.from_numpy(x[sl]).cuda() for x in gts])))
        torch.cuda.synchronize()
        singles.append((m.grads.clone(), m.moving.clone(), loss))
    gsum = singles[0][0] + singles[1][0]
    ref = YoloV3(2 * n, [img, img, 3], K, anchors, learning_rate=1e-3)
    ref.set_weights(params)
    ref.grads.copy_(gsum)
    ref.iterations = 1
    ref.lr_t_dev.fill_(ref._lr_t())
    ref._adam(ref._stream())
    torch.cuda.synchronize()
    runs = {}
    for ws in ('1', '0'):
        d = tmp_path / ('ws' + ws)
        d.mkdir()
        runs[ws] = _run_two_ranks(str(d), img, n, seed, ws)
    for ws, ranks in runs.items():
        assert int(ranks[0]['buckets']) >= 2                   # the arena really is reduced in several buckets
        for r, z in enumerate(ranks):
            assert np.array_equal(z['grads'], gsum.cpu().numpy()), 'rank %d (wgrad stream %s): all-reduced gradients != g0 + g1' % (r, ws)
            assert np.array_equal(z['params'], ref.params.cpu().numpy()), 'rank %d: weights after Adam' % r
            assert np.array_equal(z['moving'], singles[r][1].cpu().numpy()), 'rank %d: BN moving statistics must stay per replica' % r
            assert abs(float(z['loss']) - (singles[0][2] + singles[1][2])) <= 1e-6 * abs(singles[0][2] + singles[1][2])
    for r in range(2):
        for k in ('grads', 'params', 'moving'):
            assert np.array_equal(runs['1'][r][k], runs['0'][r][k]), (r, k)
    # oracle: per shard (its own batch statistics), loss / global batch, gradients summed
    tot = {}
    for dt in (torch.float32, torch.float64):
        acc = None
        for r in range(2):
            sl = slice(r * n, (r + 1) * n)
            net = om.Net(params, 3, len(anchors), K, dtype=dt, requires_grad=True)
            res = om.train_step(net, om.AdamState(net.trainable(), 1e-3), images[sl].to(dt), [torch.from_numpy(x[sl]) for x in gts],
                                (img, img, 3), anchors, K, 2 * n, apply=False)
            gs = [g.numpy().astype(np.float64) for g in res['grads']]
            acc = gs if acc is None else [a + b for a, b in zip(acc, gs)]
        tot[dt] = acc
    ref.grads.copy_(torch.from_numpy(runs['1'][0]['grads']).cuda())
    flat = []
    for sp, d in zip(ref.specs, ref.get_gradients()):
        flat += [d['W'], d['b']] + ([d['gamma'], d['beta']] if sp.bn else [])
    for i, (g, a, b) in enumerate(zip(flat, tot[torch.float32], tot[torch.float64])):
        g = np.asarray(g, np.float64)
        nb = np.linalg.norm(b) + 1e-30
        noise, err = np.linalg.norm(a - b) / nb, np.linalg.norm(g - b) / nb
        assert err <= 6.0 * noise + 5e-3, 'summed gradient tensor %d: rel L2 err %.3e (oracle fp32 noise %.3e)' % (i, err, noise)


def _run_one_rank(out_dir, img, n, seed, wgrad_stream, backend, transport):
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               Y3_WGRAD_STREAM=wgrad_stream, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    log = os.path.join(out_dir, 'rank0.log')
    with open(log, 'w') as fh:
        rc = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'dp_worker.py'), out_dir, str(img), str(n), str(seed), backend, transport],
                            env=env, stdout=fh, stderr=subprocess.STDOUT, timeout=900).returncode
    assert rc == 0, open(log).read()[-4000:]
    return np.load(os.path.join(out_dir, 'rank0.npz'))


@pytest.mark.parametrize('transport', ['torch', 'native'])
def test_rccl_backend_one_rank_forced_collectives_equal_plain_step(tmp_path, transport):
    """The REAL transport on the one GPU this box has: torch.distributed backend ``nccl`` (= RCCL) with a one-rank
    communicator and DataParallel(force_collective=True), so that every gradient bucket is all-reduced although the sum
    over one rank is the identity.  What this exercises and the gloo tests cannot: RCCL's own stream, Work.wait() as a
    STREAM wait (gloo's blocks the host and so hides any missing stream dependency), and the side-stream -> compute stream
    -> comm stream -> compute stream event chain of _Plan._run / DataParallel.  Three consecutive steps; gradients, weights,
    Adam moments and BatchNorm moving statistics must equal three plain (no process group) steps BIT FOR BIT, with the
    kernel gradients on the second stream (Y3_WGRAD_STREAM=1) and on the compute stream (=0).  ``native``: the same through
    y3_comm_* (RCCL called directly on DataParallel's comm stream)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from dp_worker import make_case
    from yolo3.model import YoloV3
    img, n, seed = 96, 4, 13
    anchors, K, params, images, gts = make_case(img, n, seed)
    m = YoloV3(n, [img, img, 3], K, anchors, learning_rate=1e-3)
    m.set_weights(params)
    for _ in range(3):
        loss = float(m.train_step((images.cuda(), [torch.from_numpy(x).cuda() for x in gts])))
    torch.cuda.synchronize()
    want = dict(grads=m.grads.cpu().numpy(), params=m.params.cpu().numpy(), moving=m.moving.cpu().numpy(), adam_m=m.adam_m.cpu().numpy())
    for ws in ('1', '0'):
        d = tmp_path / ('ws' + ws)
        d.mkdir()
        z = _run_one_rank(str(d), img, n, seed, ws, 'nccl', transport)
        assert int(z['buckets']) >= 2 and int(z['collectives']) == int(z['buckets'])      # every bucket really was all-reduced
        assert int(z['communicator_ranks']) == 1 and int(z['ranks_summed']) == 1      # ones through the gradient path (RCCL's stream)
        for k, v in want.items():
            assert np.array_equal(z[k], v), 'wgrad stream %s, %s transport: %s differs from the plain step' % (ws, transport, k)
        assert abs(float(z['loss']) - loss) <= 1e-6 * abs(loss)


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` (no launcher) spawns its own ranks and prints one JSON line (VERDICT r1 missing #1)."""
    import json
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--backend', 'gloo',
                        '--check-replicas', '--bucket-mb', '16'], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 16 and out['replicas_identical'] is True
    comm = out['comm']          # what the communicator saw and how much of the all-reduce the compute stream waited for
    assert comm['backend'] == 'gloo' and comm['world_size'] == 2 and comm['communicator_ranks'] == 2 and comm['buckets'] >= 2
    assert comm['ranks_summed'] == 2                     # ones through the gradient path: what the transport really summed over
    # gloo waits on the HOST: no event spans for the collectives, the host wait is the figure; both protocols are labelled
    assert comm['stats_protocol'] == 'own_stream' and comm['host_wait_ms'] >= 0 and comm['allreduce_ms_exposed'] >= 0
    assert comm['timed_topology']['stats_protocol'] == 'timed' and comm['timed_topology']['allreduce_ms_exposed'] >= 0
    assert comm['host']['torch_threads_per_rank'] >= 1 and out['host_issue_ms_one_step_empty_queue'] > 0
    assert np.isfinite(out['final_loss']) and out['value'] > 0 and out['scaling'] == 'weak'


def test_train_cli_two_ranks(tmp_path):
    """train.py under a two-rank launcher for one epoch (gloo: the ranks share the GPU): warm-up, N+1 steps, the
    sharded test reader, the collective checkpoint path (mean of the replicas' BN moving statistics written by rank 0
    while every replica keeps its own), export, clean shutdown of both ranks."""
    from test_gpu_cli import _write_dataset
    tmp = str(tmp_path)
    _write_dataset(tmp, 8, (160, 160, 3))
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + os.environ.get('PYTHONPATH', ''))
    out = os.path.join(tmp, 'out')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                        '--master-port', str(_free_port()), os.path.join(PKG, 'train.py'), '--batch_size', '2', '--test_every_n_steps', '2',
                        '--train_database', os.path.join(tmp, 'train-syn.lmdb'), '--test_database', os.path.join(tmp, 'test-syn.lmdb'),
                        '--output_dir', out, '--early_stopping', '1', '--use_augmentation', '0', '--max_epochs', '1', '--reader_count', '1',
                        '--backend', 'gloo'], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count('Performing Adam Optimizer learning rate warmup for 2 steps') == 2      # both ranks
    assert r.stdout.count('Train Epoch 0: Batch') == 2 * 3                                        # N+1 steps each (Q16)
    assert r.stdout.count('Test loss improved') == 2
    losses = [float(v) for v in open(os.path.join(out, 'test_loss.csv')).read().split()]
    assert len(losses) == 1 and np.isfinite(losses[0])
    from yolo3.model import YoloV3
    y = YoloV3.from_file(os.path.join(out, 'saved_model', 'yolov3.npz'))
    assert y.global_batch_size == 4 and y.img_size == [160, 160, 3]
    mov = y.moving.cpu().numpy()
    assert np.isfinite(mov).all() and not np.array_equal(mov[:y.moving_stride], np.zeros(y.moving_stride, np.float32))
