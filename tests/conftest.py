"""pytest configuration.

* `gpu` marker: tests that need a real MI355X (run by the driver with `-m gpu`; they go through
  libconvnet_hip.so and fail loudly if it is missing).
* Without a GPU (the build container) the kernel-logic tests run the SAME kernel sources through
  the TEST-ONLY SIMT emulator (libconvnet_emul.so): CONVNET_AMD_EMULATE=1 is set here and only
  here; the product never sets it.
"""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HAS_GPU = torch.cuda.is_available()
if not HAS_GPU:
    os.environ['CONVNET_AMD_EMULATE'] = '1'
    emul = os.path.join(ROOT, 'convnet.pytorch_amd', 'libconvnet_emul.so')
    srcdir = os.path.join(ROOT, 'convnet.pytorch_amd', 'csrc')
    newest = max(os.path.getmtime(os.path.join(srcdir, f)) for f in os.listdir(srcdir))
    if not os.path.exists(emul) or os.path.getmtime(emul) < newest:
        subprocess.check_call([os.path.join(srcdir, 'build.sh'), 'emul-only'])


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X GPU (HIP library, no emulator)')


def pytest_collection_modifyitems(config, items):
    for item in items:
        if 'gpu' in item.keywords and not HAS_GPU:
            item.add_marker(pytest.mark.skip(reason='no GPU visible'))


@pytest.fixture(scope='session')
def device():
    return torch.device('cuda', 0) if HAS_GPU else torch.device('cpu')
