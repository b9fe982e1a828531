"""End-to-end GPU parity of yolo3.model.YoloV3 (HIP kernels through the C ABI)
against the CPU oracle (oracle/model.py, the restatement of the reference's
model.py) on identical weights and inputs.

Tolerance rule: the network is 75 fp32 layers deep, so the yardstick is the
oracle itself -- |gpu - oracle_fp64| must stay within a small multiple of
|oracle_fp32 - oracle_fp64| (the rounding noise any fp32 implementation of the
same graph shows), plus 1e-5 of the tensor's scale.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# Absolute part of the model-level gradient bound  err <= 6 x (oracle fp32-vs-fp64 rel. L2) + GRAD_FLOOR.  What the runs measure
# (gpurun_out/train_step_grad_err_*.json, round 3, 294 tensors): at 416 x 8 the largest rel. L2 error is 1.5e-2 against an oracle
# fp32 noise of up to 1.6e-2 (medians 1.0e-2 / 1.1e-2) and the largest excess over 6 x noise is 2.1e-3; at 96 x 4: 6.6e-2 / 3.8e-2,
# excess 6.7e-4.  The floor is ~2.4 x the largest excess.  The per-kernel tests (2e-5 .. 1e-4 against fp64) carry the real weight:
# this test checks that the 370-launch plan wires them into the right graph.
GRAD_FLOOR = 5e-3 * (1.5 if os.environ.get('Y3_NO_FAST') else 1.0)   # generic kernel (Y3_NO_FAST=1): another summation order, another set of flips; largest excess seen 5.3e-3
# conv_arithmetic 'x3' (the default since round 4): yet another rounding pattern, yet another set of flips.  Measured at 96 x 4: one
# tensor (the beta gradient of a 576-pixel layer: a single leaky-relu flip is 5e-3 of it) at 5.76e-3 against an oracle noise of
# 3.9e-5, i.e. an excess of 5.5e-3; every x3 kernel is closer to fp64 than its fp32-MFMA twin in isolation
# (test_conv_x3_error_against_fp64_is_that_of_the_f32_instruction), so the floor is a statement about flips, not about arithmetic.
# tests/grad_err_report.py (profiles/r04_grad_err_96_4.txt) shows the whole distribution at 96 x 4: against the fp64 oracle the HIP x3 step
# is at a median rel. L2 of 4.5e-3 (max 1.8e-2) where the torch-CPU fp32 oracle itself is at 1.7e-2 (max 3.8e-2) and the HIP
# fp32-MFMA step at 3.7e-2 (max 6.6e-2): x3 is the CLOSEST of the three to fp64 (one rounding per 16 exact products), so its
# errors do not line up tensor by tensor with the fp32 oracle's noise the way another fp32 evaluation's do.
GRAD_FLOOR_X3 = 1e-2
ANCHORS = [(64, 384), (384, 64)]
K = 2


def _bound(ref32, ref64, scale_floor=1e-30, mult=6.0, rel=1e-5):
    noise = float(np.abs(np.asarray(ref32, np.float64) - np.asarray(ref64, np.float64)).max())
    scale = max(float(np.abs(np.asarray(ref64)).max()), scale_floor)
    return mult * noise + rel * scale


def _check(got, ref32, ref64, what, **kw):
    got = np.asarray(got, np.float64)
    ref64 = np.asarray(ref64, np.float64)
    assert got.shape == ref64.shape, (what, got.shape, ref64.shape)
    assert np.isfinite(got).all(), what
    err = float(np.abs(got - ref64).max())
    b = _bound(ref32, ref64, **kw)
    assert err <= b, '%s: max err %.3e > bound %.3e' % (what, err, b)


def _setup(img, n, seed, randomize_bn, conv_arithmetic=None):
    from oracle import model as om
    from yolo3.model import YoloV3
    from test_gpu_kernels import _labels
    params = om.init_params(3, len(ANCHORS), K, seed=seed, randomize_bn=randomize_bn)
    if randomize_bn:
        # randomised BN makes activations grow with depth; keep the head logits O(1) so exp() in the decode / loss
        # stays inside fp32 range (otherwise fp32 and fp64 evaluations of the SAME graph disagree by overflow)
        for p in params:
            if 'gamma' not in p:
                p['W'] *= 0.02
    yolo = YoloV3(n, [img, img, 3], K, ANCHORS, learning_rate=1e-3, conv_arithmetic=conv_arithmetic)
    yolo.set_weights(params)
    g = torch.Generator().manual_seed(seed)
    images = torch.randn(n, 3, img, img, generator=g)
    gts = _labels(np.random.default_rng(seed), n, img, ANCHORS, K, per_image=3)
    return om, params, yolo, images, gts


def test_weight_preparation_in_one_pass_equals_the_three_launches(monkeypatch):
    """y3_x3_prepare_weights_batched (transposed copy + piece planes of both copies in one pass over the arena) writes, bit for bit,
    what y3_transpose_weights_batched and the two y3_x3_split_weights_batched launches write."""
    om, params, yolo, _, _ = _setup(64, 1, 5, True)
    assert yolo.planes is not None
    outs = []
    for fused in ('0', '1'):
        monkeypatch.setenv('Y3_PREP_FUSED', fused)
        for t in (yolo.params_t, yolo.planes, yolo.planes_t):
            t.zero_()
        yolo._refresh_transposed()
        torch.cuda.synchronize()
        outs.append([yolo.params_t.clone(), yolo.planes.view(torch.int16).clone(), yolo.planes_t.view(torch.int16).clone()])
    for a, b, name in zip(outs[0], outs[1], ('params_t', 'planes', 'planes_t')):
        assert torch.equal(a, b), name
    assert int((outs[1][1] != 0).sum()) > 1000000 and int((outs[1][2] != 0).sum()) > 1000000


def test_set_get_weights_roundtrip():
    om, params, yolo, _, _ = _setup(64, 1, 3, True)
    back = yolo.get_weights()
    for a, b in zip(params, back):
        for k in a:
            assert np.array_equal(a[k], b[k]), k
    assert sum(int(np.prod(w.shape)) for w in yolo.trainable_weights()) == 61789770
    assert len(yolo.trainable_weights()) == 294


@pytest.mark.parametrize('img,n', [(96, 3), (416, 1), (416, 8)])
def test_inference_matches_oracle(img, n):
    """predict(): conv stacks with folded BN + residuals + upsample/concat + decode.  (416, 8) is BASELINE.json
    configs[1] at its full size: the launch planner picks other tiles / split-K factors there than at n = 1."""
    om, params, yolo, images, _ = _setup(img, n, 7, True)
    out = yolo.predict(images.cuda()).cpu().numpy()
    refs = {}
    for dt in (torch.float32, torch.float64):
        net = om.Net(params, 3, len(ANCHORS), K, dtype=dt)
        with torch.no_grad():
            fms = net.feature_maps(images.to(dt), training=False)
            refs[dt] = (om.decode(fms, (img, img, 3), ANCHORS, K).numpy(), [f.numpy() for f in fms])
    nb = 2 * sum((img // s) ** 2 for s in (32, 16, 8))
    assert out.shape == (n, nb, 5 + K)
    fm_gpu = [f.cpu().numpy() for f in yolo.feature_maps(images.cuda(), training=False)]
    for i in range(3):
        _check(fm_gpu[i], refs[torch.float32][1][i], refs[torch.float64][1][i], 'feature_map_%d' % (i + 1))
    _check(out[..., 4:], refs[torch.float32][0][..., 4:], refs[torch.float64][0][..., 4:], 'scores')
    # boxes: exp() turns logit noise into relative noise, so compare relative to the box magnitude
    b32, b64 = refs[torch.float32][0][..., :4], refs[torch.float64][0][..., :4]
    den = np.abs(b64) + 1.0
    _check(out[..., :4] / den, b32 / den, b64 / den, 'boxes', mult=10.0)


def _rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


@pytest.mark.parametrize('img,n', [(96, 3), (416, 1)])
def test_bf16_inference_matches_bf16_oracle(img, n):
    """predict(precision='bf16') (BASELINE config 5's conv path) against the oracle run with the SAME rounding points:
    bf16 kernels (layers 2..75), one bf16 rounding per stored activation, fp32 accumulation / BN fold / heads / decode.
    Products of bf16 pairs are exact in fp32, so the two differ only by fp32 summation order -- which can flip a
    round-to-nearest decision (1 bf16 ulp = 2^-8 on that element).  Yardstick: the distance to the emulation must be
    no larger than the bf16 path's own distance to the fp32 network (end to end the flips compound through 75 layers;
    the sharp per-layer statement is test_bf16_layers_teacher_forced below)."""
    om, params, yolo, images, _ = _setup(img, n, 7, True)
    out16 = yolo.predict(images.cuda(), precision='bf16').cpu().numpy()
    fm16 = [f.cpu().numpy() for f in yolo.feature_maps(images.cuda(), precision='bf16')]
    out32 = yolo.predict(images.cuda()).cpu().numpy()                 # the fp32 plan is untouched by the bf16 one
    net = om.Net(params, 3, len(ANCHORS), K, dtype=torch.float32)
    with torch.no_grad():
        fms32_t = net.feature_maps(images, training=False)
        dec_32 = om.decode(fms32_t, (img, img, 3), ANCHORS, K).numpy()
        fms32 = [f.numpy() for f in fms32_t]
        net.bf16 = True
        fms_e = net.feature_maps(images, training=False)
        dec_e = om.decode(fms_e, (img, img, 3), ANCHORS, K).numpy()
        fms_e = [f.numpy() for f in fms_e]
    assert np.isfinite(out16).all() and out16.shape == out32.shape
    for i in range(3):
        d_emul = _rel_l2(fm16[i], fms_e[i])
        d_prec = _rel_l2(fms_e[i], fms32[i])
        print('fm%d  bf16-kernel vs bf16-oracle %.3e   bf16-oracle vs fp32-oracle %.3e' % (i + 1, d_emul, d_prec))
        assert d_emul <= 2.0 * d_prec + 1e-4, (i, d_emul, d_prec)
        assert d_prec < 0.05                                      # bf16 itself stays a faithful approximation here
    # decoded rows (exp() of the wh logits amplifies the flips): the same yardstick -- no farther from the emulation than twice the
    # emulation's own distance to the fp32 network -- and small in absolute terms (measured at 416: 2.0e-2)
    d_emul, d_prec = _rel_l2(out16, dec_e), _rel_l2(dec_e, dec_32)
    print('rows bf16-kernel vs bf16-oracle %.3e   bf16-oracle vs fp32-oracle %.3e' % (d_emul, d_prec))
    assert d_emul <= 2.0 * d_prec + 1e-4 and d_emul <= 4e-2, (d_emul, d_prec)
    yolo2 = yolo.__class__(n, [img, img, 3], K, ANCHORS, inference_precision='bf16')
    yolo2.set_weights(params)
    assert torch.equal(yolo2.predict(images.cuda()).cpu(), torch.from_numpy(out16))    # constructor default, deterministic


@pytest.mark.parametrize('img,n', [(96, 2), (608, 2)])
def test_bf16_layers_teacher_forced(img, n):
    """(608 = the tile size of BASELINE.json configs[4]: the real model on a batch of full-size tiles.)
    Every layer of the bf16 plan on its own: the oracle (fp64 arithmetic, bf16 rounding points) recomputes layer i
    from the GPU's OWN bf16 inputs (teacher forcing), so nothing compounds.  Bound per element against the oracle's
    value BEFORE rounding: half a bf16 ulp (round to nearest; a tie or near-tie may legitimately fall either way, which
    is why the comparison is not against the oracle's rounded value) plus 3e-5 of the layer's scale for fp32
    accumulation order / the folded BatchNorm affine."""
    om, params, yolo, images, _ = _setup(img, n, 11, True)
    yolo.predict(images.cuda(), precision='bf16')
    plan = yolo._plan(n, False, True)
    torch.cuda.synchronize()
    nchw = lambda t: t.torch_view().float().permute(0, 3, 1, 2).cpu()
    layers = [nchw(t) for t in plan.layer_out]
    ups = [nchw(op[2]) for op in plan.ops if op[0] == 'upsample']
    fms_gpu = [f.cpu() for f in yolo.feature_maps(images.cuda(), precision='bf16')]
    assert len(layers) == 72 and len(ups) == 2
    net = om.Net(params, 3, len(ANCHORS), K, dtype=torch.float64)
    net.bf16 = True
    net.trace_exact, net.up_trace = [], []
    net.force = {'layers': layers, 'up': ups}
    with torch.no_grad():
        fms = net.feature_maps(images.double(), training=False)

    def ok(got, ref, what):
        got, ref = got.double(), ref.double()
        ulp = torch.exp2(torch.floor(torch.log2(ref.abs().clamp_min(1e-30))) - 7)      # bf16: 8 significant bits
        bound = 0.5 * ulp + 3e-5 * float(ref.abs().max())
        bad = (got - ref).abs() > bound
        assert torch.isfinite(got).all() and not bool(bad.any()), '%s: %d elements off, worst excess %.3e' % (
            what, int(bad.sum()), float(((got - ref).abs() - bound).max()))

    assert len(net.trace_exact) == 72
    for j, (g, r) in enumerate(zip(layers, net.trace_exact)):
        ok(g, r, 'conv_layer %d' % j)
    for j, (g, r) in enumerate(zip(ups, net.up_trace)):
        ok(g[:, :r.shape[1]], r, 'upsample %d' % j)
    for j, (g, r) in enumerate(zip(fms_gpu, fms)):
        assert_fm = (g.double() - r).abs().max() <= 3e-5 * float(r.abs().max())     # fp32 heads: no rounding at all
        assert bool(assert_fm), 'head %d' % j


_ORACLE_STEPS = {}


@pytest.mark.parametrize('img,n,arith', [(96, 4, 'x3'), (96, 4, 'f32'), (416, 8, 'x3')])
def test_train_step_matches_oracle(img, n, arith):
    """train_step(): forward with batch statistics, loss, full backward (dgrad/wgrad/BN/upsample), Keras Adam,
    moving-stat update.  Step 1 is compared tensor by tensor; step 2 only through its loss, because the first Adam
    step moves every weight by ~lr*sign(g) and the sign of a numerically-zero gradient is implementation noise.
    (416, 8) is the benchmarked step (BASELINE.json configs[2]) at its full size, in the arithmetic bench.py times (x3 where the
    model's policy uses it); (96, 4) runs both arithmetics.
    HOW SHARP THIS IS: the gradient bound is 6 x the oracle's OWN fp32-vs-fp64 relative L2 distance + a floor.  At 416 x 8 that
    distance is up to 1.6e-2 per tensor (median 1.1e-2) and the HIP step measures up to 1.5e-2 (round 3,
    gpurun_out/train_step_grad_err_416_8.json): this is a 1e-2-class statement about the wiring of ~370 launches, NOT a 1e-5
    statement about arithmetic -- it cannot see a 1 % error in one tensor.  The per-kernel tests of test_gpu_kernels.py
    (2e-5 .. 1e-4 against fp64) are the ones that bound the arithmetic."""
    om, params, yolo, images, gts = _setup(img, n, 11, False, conv_arithmetic=arith)
    floor = GRAD_FLOOR_X3 if arith == 'x3' else GRAD_FLOOR
    gbs = n
    res = _ORACLE_STEPS.get((img, n))          # the oracle's two steps do not depend on the arithmetic under test: computed once per size
    if res is None:
        res = {}
        for dt in (torch.float32, torch.float64):
            net = om.Net(params, 3, len(ANCHORS), K, dtype=dt, requires_grad=True)
            adam = om.AdamState(net.trainable(), 1e-3)
            steps, snaps = [], []
            for _ in range(2):
                steps.append(om.train_step(net, adam, images.to(dt), [torch.from_numpy(g) for g in gts], (img, img, 3), ANCHORS, K, gbs))
                snaps.append(([t.detach().numpy().copy() for t in net.trainable()],
                              [(q['mean'].numpy().copy(), q['var'].numpy().copy()) for q in net.p if 'mean' in q]))
            res[dt] = (steps, snaps)
        if img <= 96:
            _ORACLE_STEPS[(img, n)] = res
    gt_dev = [torch.from_numpy(g).cuda() for g in gts]
    from yolo3.model import Mean
    mets = [Mean() for _ in range(5)]
    r32, r64 = res[torch.float32][0][0], res[torch.float64][0][0]
    loss = yolo.train_step((images.cuda(), gt_dev, *mets))
    assert abs(float(loss) - r64['loss']) <= 6 * abs(r32['loss'] - r64['loss']) + 1e-5 * abs(r64['loss']), (float(loss), r64['loss'])
    np.testing.assert_allclose([m.result() for m in mets], [r64['loss']] + r64['parts'], rtol=1e-4)
    flat = []
    for sp, d in zip(yolo.specs, yolo.get_gradients()):
        flat += [d['W'], d['b']] + ([d['gamma'], d['beta']] if sp.bn else [])
    # The backward pass is discontinuous in its inputs (leaky-relu slope at a ~ 0, the ignore mask at IoU ~ 0.5):
    # two fp32 evaluations of the same graph differ by a few per cent in max-norm on tensors with few pixels
    # (the fp32 and fp64 oracles do, too).  Compare in relative L2 against the oracle's own fp32-vs-fp64 gap;
    # each kernel is checked to 1e-4..2e-5 in isolation by test_gpu_kernels.py.
    report = []
    for i, (g, a, b) in enumerate(zip(flat, r32['grads'], r64['grads'])):
        a, b, g = a.numpy().astype(np.float64), b.numpy(), np.asarray(g, np.float64)
        nb = np.linalg.norm(b) + 1e-30
        noise, err = np.linalg.norm(a - b) / nb, np.linalg.norm(g - b) / nb
        report.append((err, noise))
        assert np.isfinite(g).all() and err <= 6.0 * noise + floor, 'grad tensor %d: rel L2 err %.3e (oracle fp32 noise %.3e)' % (i, err, noise)
    out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out')
    if os.path.isdir(out_dir):      # what this run measured (the floor above is 2 x the largest excess seen: see GRAD_FLOOR)
        import json
        errs, noises = np.array([r[0] for r in report]), np.array([r[1] for r in report])
        with open(os.path.join(out_dir, 'train_step_grad_err_%d_%d_%s.json' % (img, n, arith)), 'w') as fh:
            json.dump(dict(tensors=len(report), conv_arithmetic=arith, max_rel_l2_err=float(errs.max()), median_rel_l2_err=float(np.median(errs)),
                           max_oracle_fp32_noise=float(noises.max()), median_oracle_fp32_noise=float(np.median(noises)),
                           max_excess_over_6x_noise=float((errs - 6.0 * noises).max())), fh)
    # moving statistics after the first forward depend on the initial weights only
    mov = [(d['mean'], d['var']) for d in yolo.get_weights() if 'mean' in d]
    for i, ((m, v), (m32, v32), (m64, v64)) in enumerate(zip(mov, res[torch.float32][1][0][1], res[torch.float64][1][0][1])):
        _check(m, m32, m64, 'moving mean %d' % i, rel=1e-4)
        _check(v, v32, v64, 'moving var %d' % i, rel=1e-4)
    # weights after one Adam step: |delta| <= lr everywhere; they agree except where the gradient is numerically zero
    for i, (w, a, b) in enumerate(zip(yolo.trainable_weights(), res[torch.float32][1][0][0], res[torch.float64][1][0][0])):
        d = np.abs(np.asarray(w, np.float64) - b)
        assert d.max() <= 2.05e-3, i
        assert np.median(d) <= max(4 * np.median(np.abs(a - b)), 1e-6), i
    loss2 = float(yolo.train_step((images.cuda(), gt_dev)))
    r64b = res[torch.float64][0][1]['loss']
    assert np.isfinite(loss2) and abs(loss2 - r64b) <= 0.05 * abs(r64b), (loss2, r64b)


def test_test_step_and_graph_replay():
    """test_step (BN moving stats, no update) and the HIP-graph path of train_step give the eager results."""
    img, n = 64, 2
    om, params, yolo, images, gts = _setup(img, n, 13, True)
    net = om.Net(params, 3, len(ANCHORS), K, dtype=torch.float64)
    want, parts = om.test_step(net, images.double(), [torch.from_numpy(g) for g in gts], (img, img, 3), ANCHORS, K, n)
    gt_dev = [torch.from_numpy(g).cuda() for g in gts]
    got = float(yolo.test_step((images.cuda(), gt_dev)))
    assert abs(got - want) <= 2e-4 * abs(want), (got, want)
    # eager vs graph: same weights, same data -> same losses step by step
    from yolo3.model import YoloV3
    a = YoloV3(n, [img, img, 3], K, ANCHORS, learning_rate=1e-3)
    b = YoloV3(n, [img, img, 3], K, ANCHORS, learning_rate=1e-3, use_graph=True)
    a.set_weights(params)
    b.set_weights(params)
    # a launches from the host with the kernel gradients on a second stream, b replays a single-stream graph: the kernels
    # and their accumulation orders are the same, so gradients and weights must agree BIT FOR BIT -- any hazard between
    # the two streams (a shared workspace, a buffer reused too early) shows up here
    la, lb = [], []
    for _ in range(4):
        la.append(float(a.train_step((images.cuda(), gt_dev))))
        lb.append(float(b.train_step((images.cuda(), gt_dev))))
        torch.cuda.synchronize()
        assert torch.equal(a.grads, b.grads) and torch.equal(a.params, b.params) and torch.equal(a.moving, b.moving)
    np.testing.assert_allclose(la, lb, rtol=0)
    for wa, wb in zip(a.trainable_weights(), b.trainable_weights()):
        np.testing.assert_allclose(wa, wb, rtol=0, atol=2.1e-3)
    # inference graphs (fp32 and bf16 plans) replay to the eager rows, before and after the weights move again
    for prec in ('fp32', 'bf16'):
        b.set_weights(a.get_weights())
        for _ in range(2):
            ea = a.predict(images.cuda(), precision=prec).clone()
            gb = b.predict(images.cuda(), precision=prec).clone()
            assert torch.equal(ea, gb), prec
            a.train_step((images.cuda(), gt_dev))
            b.set_weights(a.get_weights())


def test_graph_replay_after_an_idle_gpu():
    """Regression (round 3): the loss entry cleared its anchor flags with hipMemsetAsync; as a memset NODE of the captured training
    step that clear was not reliably ordered against the loss kernels of the previous scale, and a replay that followed a period
    of idle GPU (here: another model's eager step with Y3_CHECK_TICKETS=1, which synchronises before every ticketed launch)
    computed a wrong loss while every BatchNorm statistic was still right.  The clear is a kernel now.  In a process of its own:
    the switch is read once per process."""
    import subprocess
    import sys
    code = """
import sys, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import test_gpu_model as T
from yolo3.model import YoloV3
om, params, yolo, images, gts = T._setup(64, 2, 13, True)
gt = [torch.from_numpy(g).cuda() for g in gts]
mk = lambda g: YoloV3(2, [64, 64, 3], T.K, T.ANCHORS, learning_rate=1e-3, use_graph=g)
a, b = mk(False), mk(True)
a.set_weights(params); b.set_weights(params)
for s in range(4):
    la = float(a.train_step((images.cuda(), gt)))
    lb = float(b.train_step((images.cuda(), gt)))
    torch.cuda.synchronize()
    assert la == lb and torch.equal(a.grads, b.grads) and torch.equal(a.params, b.params), (s, la, lb)
print('ok')
""" % (os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'object-detection-yolov3_amd'), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, Y3_CHECK_TICKETS='1')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'ok' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_nonsquare_grayscale_three_anchors():
    """Edge cases of the contract: 1-channel input (padded 1 -> 4 internally), non-square image (the Q6 stride quirk:
    x uses H//G_h, y uses W//G_w), the default three anchors and three classes (24 head channels), odd grid sizes.
    Inference and one training step against the oracle."""
    from oracle import model as om
    from yolo3.model import YoloV3
    from yolo3.imagereader import format_boxes
    anchors, K3, H, W, n = [(32, 32), (128, 128), (256, 256)], 3, 96, 160, 2
    params = om.init_params(1, 3, K3, seed=21, randomize_bn=True)
    for p in params:
        if 'gamma' not in p:
            p['W'] *= 0.02
    g = torch.Generator().manual_seed(21)
    images = torch.randn(n, 1, H, W, generator=g)
    yolo = YoloV3(n, [H, W, 1], K3)             # anchors=None -> the reference default (model.py:433)
    assert yolo.anchors == anchors
    yolo.set_weights(params)
    out = yolo.predict(images.cuda()).cpu().numpy()
    refs = {}
    for dt in (torch.float32, torch.float64):
        net = om.Net(params, 1, 3, K3, dtype=dt)
        with torch.no_grad():
            refs[dt] = om.decode(net.feature_maps(images.to(dt), training=False), (H, W, 1), anchors, K3).numpy()
    assert out.shape == (n, 3 * (3 * 5 + 6 * 10 + 12 * 20), 5 + K3)
    den = np.abs(refs[torch.float64]) + 1.0
    _check(out / den, refs[torch.float32] / den, refs[torch.float64] / den, 'non-square decode', mult=10.0)
    # one training step (Keras-default BN so batch statistics matter)
    params = om.init_params(1, 3, K3, seed=22)
    yolo.set_weights(params)
    rng = np.random.default_rng(22)
    labs = []
    for _ in range(n):
        wh = rng.integers(20, 60, (2, 2))
        xy = np.stack([rng.integers(0, W - wh[:, 0]), rng.integers(0, H - wh[:, 1])], 1)
        labs.append(format_boxes(np.concatenate([xy, wh, rng.integers(0, K3, (2, 1))], 1).astype(np.int32), (H, W, 1), anchors, K3))
    gts = [np.stack([l[s] for l in labs]) for s in range(3)]
    res = {}
    for dt in (torch.float32, torch.float64):
        net = om.Net(params, 1, 3, K3, dtype=dt, requires_grad=True)
        res[dt] = om.train_step(net, om.AdamState(net.trainable(), 1e-3), images.to(dt), [torch.from_numpy(x) for x in gts], (H, W, 1), anchors, K3, n, apply=False)
    loss = float(yolo.train_step((images.cuda(), [torch.from_numpy(x).cuda() for x in gts])))
    r32, r64 = res[torch.float32], res[torch.float64]
    assert abs(loss - r64['loss']) <= 6 * abs(r32['loss'] - r64['loss']) + 1e-5 * abs(r64['loss']), (loss, r64['loss'])
    flat = []
    for sp, d in zip(yolo.specs, yolo.get_gradients()):
        flat += [d['W'], d['b']] + ([d['gamma'], d['beta']] if sp.bn else [])
    for i, (gg, a, b) in enumerate(zip(flat, r32['grads'], r64['grads'])):
        a, b, gg = a.numpy().astype(np.float64), b.numpy(), np.asarray(gg, np.float64)
        nb = np.linalg.norm(b) + 1e-30
        # (default arithmetic: x3; 48 x 80 images: tensors of 60-240 pixels per channel, where ONE flipped leaky-relu slope is 1e-2 of a bias gradient: measured 1.3e-2)
        assert np.isfinite(gg).all() and np.linalg.norm(gg - b) / nb <= 6.0 * np.linalg.norm(a - b) / nb + 2 * GRAD_FLOOR_X3, i
