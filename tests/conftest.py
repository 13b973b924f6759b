import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for path in (ROOT, os.path.join(ROOT, 'oracle')):
    if path not in sys.path:
        sys.path.insert(0, path)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'))


@pytest.fixture(scope='session')
def golden():
    return load_golden


_dev_library = None


def dev_library():
    """libtonic_hip_dev.so (`make -C tonic_amd/csrc dev`): the product sources with -DTONIC_DEV, i.e. plus the
    REFERENCE forms of the fused PPO grad kernels (grad_variant 0 - 3: fp32 MFMA, bf16x3) that the shipped
    fp16x2 form is compared with.  Test scaffolding: nothing in tonic_amd/ or bench.py loads it."""
    global _dev_library
    if _dev_library is None:
        import ctypes
        from tonic_amd import _lib
        path = os.path.join(ROOT, 'tonic_amd', 'libtonic_hip_dev.so')
        if not os.path.exists(path):
            pytest.skip('libtonic_hip_dev.so not built (make -C tonic_amd/csrc dev)')
        lib = ctypes.CDLL(path)
        for name, (restype, argtypes) in _lib.SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = restype, argtypes
        assert lib.tonic_abi_version() == _lib.ABI_VERSION
        _dev_library = lib
    return _dev_library


def variant_library(lib, variant):
    """The library that holds `grad_variant`: the product library for its own (4 / default), else the dev one."""
    return lib if variant in (None, -1, 4) else dev_library()
