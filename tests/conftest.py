import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'object-detection-yolov3_amd')
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    lib = os.path.join(PKG, 'yolo3', '_lib', 'libyolo3hip.so')
    if not os.path.exists(lib):   # hipcc cross-compiles without a GPU; the library is git-ignored
        subprocess.check_call(['make', '-C', os.path.join(PKG, 'csrc')])


@pytest.fixture(scope='session')
def golden_dir():
    return os.path.join(ROOT, 'tests', 'golden')
