"""The host side of the convolution planners under AddressSanitizer + UBSan.

conv.hip's and core.hip's HOST code is compiled without any device code (hipcc --cuda-host-only, seconds, no GPU) with
-fsanitize=address,undefined and linked with tests/planner_sweep.cpp, which asks for the plan of every convolution of the
network (forward, data gradient incl. the stride-2 parity classes, kernel gradient) and replays the kernels' index
arithmetic -- work item -> tile / K slice / slab slot / ticket -- against the workspace size the query functions report.
Run for batch 1 / 2 / 8 / 25 at 416 and 608 with the product's defaults, and -- in a -DY3_DEV build, where the development
switches exist at all -- with the switch combinations the probes under tools/probe use (Y3_RSPLIT, Y3_TILE,
Y3_SPLITK_WGS / _MINK, Y3_WGRAD_WAVES, Y3_WGRAD_TILE, Y3_NO_FAST)."""
import ctypes as C
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'object-detection-yolov3_amd', 'csrc')
HIPCC = '/opt/rocm/bin/hipcc'
CLANG = '/opt/rocm/lib/llvm/bin/clang++'
SAN = ['-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer']

pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(CLANG)), reason='ROCm toolchain not installed')


def _build(out_dir, dev):
    exe = os.path.join(out_dir, 'planner_sweep_dev' if dev else 'planner_sweep')
    objs = []
    for src in ('conv.hip', 'conv_x3.hip', 'core.hip'):
        obj = os.path.join(out_dir, src.replace('.hip', '_dev.o' if dev else '.o'))
        subprocess.check_call([HIPCC, '-O1', '-g', '-std=c++17', '--cuda-host-only', '-ffp-contract=off', '-Wno-unused-function'] + SAN +
                              (['-DY3_DEV'] if dev else []) + ['-I', os.path.join(ROOT, 'include'), '-c', os.path.join(CSRC, src), '-o', obj])
        objs.append(obj)
    drv = os.path.join(out_dir, 'drv.o')
    subprocess.check_call([CLANG, '-O1', '-g', '-std=c++17'] + SAN + ['-c', os.path.join(ROOT, 'tests', 'planner_sweep.cpp'), '-o', drv])
    # a host-only object still references the fat binary its device pass would have produced (__hip_fatbin_<hash>): the only
    # symbol allowed to stay unresolved -- nothing here launches a kernel
    subprocess.check_call([CLANG] + SAN + [drv] + objs + ['-L/opt/rocm/lib', '-lamdhip64', '-Wl,-rpath,/opt/rocm/lib', '-ldl',
                                                         '-Wl,--unresolved-symbols=ignore-all', '-o', exe])
    und = subprocess.run(['nm', '-u', exe], capture_output=True, text=True).stdout.split()
    stray = [s for s in und if s.startswith('y3_') or s.startswith('_Z') and 'y3' in s]
    assert not stray, 'unresolved product symbols: %s' % stray
    return exe


@pytest.fixture(scope='module')
def sweeps(tmp_path_factory):
    d = str(tmp_path_factory.mktemp('planner'))
    return _build(d, False), _build(d, True)


def _run(exe, args, env=None):
    e = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    for k in list(e):
        if k.startswith('Y3_'):
            del e[k]
    e.update(env or {})
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, env=e, timeout=300)
    assert r.returncode == 0 and 'runtime error' not in r.stderr and 'AddressSanitizer' not in r.stderr, \
        '%s %s %s\n%s\n%s' % (os.path.basename(exe), args, env, r.stdout[-1500:], r.stderr[-3000:])
    return r.stdout


CASES = [(1, 416), (2, 416), (8, 416), (2, 608), (8, 608), (25, 608), (8, 96), (4, 64)]


def test_planner_sweep_product_defaults(sweeps):
    exe, _ = sweeps
    for batch, img in CASES:
        out = _run(exe, [batch, img])
        assert out.strip().endswith('ok 75 layers batch=%d img=%d' % (batch, img))
    # one more head width (3 anchors x 8 classes = 39 channels, not a multiple of 4)
    _run(exe, [8, 416, 39])


def test_product_build_ignores_development_switches(sweeps):
    """Without -DY3_DEV the tuning switches do not exist: the plans do not move."""
    exe, _ = sweeps
    base = _run(exe, [8, 416])
    assert _run(exe, [8, 416], {'Y3_TILE': '64,64,16', 'Y3_RSPLIT': '0', 'Y3_SPLITK_WGS': '500', 'Y3_WGRAD_WAVES': '1024', 'Y3_PIPE': '1'}) == base
    assert _run(exe, [8, 416], {'Y3_X3_SLOTS': '1000', 'Y3_X3_KS': '3', 'Y3_X3_BN': '64', 'Y3_X3_RSPLIT': '0', 'Y3_WGX3_WGS': '900', 'Y3_BNB_LC': '4', 'Y3_BNB_BLOCKS': '64',
                                'Y3_X3_OVERFLOW': '0', 'Y3_KORDER': '0', 'Y3_X3_MODE': '0'}) == base
    assert _run(exe, [8, 416], {'Y3_NO_FAST': '1'}) != base      # the one switch the product reads (generic kernel everywhere)


DEV_ENVS = [
    {},
    {'Y3_PIPE': '1', 'Y3_RSPLIT': '1'},          # the combination of gpurun_out/r02_conv_timing1.log:165
    {'Y3_PIPE': '1', 'Y3_RSPLIT': '0'},
    {'Y3_PIPE': '0', 'Y3_RSPLIT': '0'},
    {'Y3_RSPLIT': '0', 'Y3_SPLITK_MINK': '100000'},
    {'Y3_TILE': '64,64,16'},
    {'Y3_TILE': '128,128,16'},
    {'Y3_TILE': '64,128,16', 'Y3_SPLITK_WGS': '2800'},
    {'Y3_TILE': '128,64,16', 'Y3_SPLITK_WGS': '1000', 'Y3_SPLITK_MINK': '64'},
    {'Y3_SPLITK_WGS': '100000', 'Y3_SPLITK_MINK': '16'},
    {'Y3_CUS': '64'},
    {'Y3_WGRAD_WAVES': '1024'},
    {'Y3_WGRAD_WAVES': '16384'},
    {'Y3_WGRAD_TILE': '64,64', 'Y3_WGRAD_TILE_MAXK': '100000'},
    {'Y3_WGRAD_TILE': '128,128', 'Y3_WGRAD_TILE_MAXK': '100000', 'Y3_WGRAD_INKERNEL': '0'},
    {'Y3_WGRAD_SHAPE_RULES': '0'},
    {'Y3_NO_FAST': '1'},
    {'Y3_NO_DGRAD_MULTI': '1'},
    {'Y3_X3_SLOTS': '768'},
    {'Y3_X3_SLOTS': '256', 'Y3_X3_RSPLIT': '0'},
    {'Y3_X3_KS': '3', 'Y3_X3_RSPLIT': '0'},
    {'Y3_X3_BN': '64', 'Y3_X3_KS': '16'},
    {'Y3_WGX3_WGS': '1500'},
    {'Y3_WGX3_WGS': '100'},
    {'Y3_X3_OVERFLOW': '0'},
    {'Y3_KORDER': '0'},
]


def test_planner_sweep_development_switches(sweeps):
    _, exe = sweeps
    for env in DEV_ENVS:
        for batch, img in ((8, 416), (2, 608), (25, 608), (1, 416)):
            _run(exe, [batch, img], env)


def test_plans_match_the_product_library(sweeps):
    """The sanitizer build and the shipped .so are the same planner: same numbers for the benchmarked configuration."""
    exe, _ = sweeps
    sys.path.insert(0, os.path.join(ROOT, 'object-detection-yolov3_amd'))
    from yolo3 import _hip
    rows = [ln.split() for ln in _run(exe, [8, 416]).splitlines() if ln.startswith(('conv ', 'wgrad ', 'wgrad_x3 '))]
    assert len(rows) > 200 and any('_x3' in r[1] for r in rows if r[0] == 'conv') and any(r[0] == 'wgrad_x3' for r in rows)
    for r in rows:
        kv = dict(x.split('=') for x in r[1:] if '=' in x)
        m, cin, k, cout = (int(kv[x]) for x in ('m', 'cin', 'k', 'cout'))
        if r[0] == 'conv':
            o = (C.c_int * 13)()
            ws = _hip.lib.y3_conv2d_plan_x(m, cin, k, cout, _hip.CONV_X3 if '_x3' in r[1] else 0, o)
            want = [int(kv[x]) for x in ('bm', 'bn', 'bk', 'tiles', 'f', 's0', 's1', 'c0', 'c1', 'grid', 'stats', 'fast', 'nk')]
        else:
            o = (C.c_int * 8)()
            ws = _hip.lib.y3_conv2d_wgrad_plan_x(m, cin, k, cout, _hip.CONV_X3 if r[0] == 'wgrad_x3' else 0, o)
            want = [int(kv[x]) for x in ('bkr', 'bn', 'splits', 'chunk', 'tiles', 'in_kernel', 'grid')]
        assert list(o)[:len(want)] == want and int(ws) == int(kv['ws']), r
