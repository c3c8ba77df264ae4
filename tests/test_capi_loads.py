"""CPU-side check of the boundary: the C-ABI library builds, loads, and exports every symbol that
include/blueice_hip.h declares (no compute calls -- there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    from blueice_amd import build, _capi
    build.build()
    return _capi.load()


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'blueice_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(bi_[a-z_0-9]+)\s*\(', text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from blueice_amd import _capi
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "library does not export %s" % s
        assert s in _capi.SIGNATURES, "ctypes binding lacks %s" % s
    assert sorted(_capi.SIGNATURES) == syms


def test_version_and_loud_failure_without_gpu(lib):
    import ctypes as C
    assert b'gfx950' in lib.bi_version()
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip('a GPU is present')
    h = C.c_void_p()
    assert lib.bi_create(0, C.byref(h)) != 0
    assert b'no CPU fallback' in lib.bi_last_error(None)
    from blueice_amd.device import DeviceContext
    from blueice_amd.exceptions import DeviceError
    with pytest.raises(DeviceError):
        DeviceContext(0)
