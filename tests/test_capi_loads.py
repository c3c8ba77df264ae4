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


def test_parameter_names_are_enumerable_and_documented(lib):
    """bi_list_params lists every tunable / counter / trigger, and the header documents each of them."""
    import ctypes as C
    n = lib.bi_list_params(None, 0)
    buf = C.create_string_buffer(n)
    assert lib.bi_list_params(buf, n) == n
    params = dict(line.split() for line in buf.value.decode().splitlines())
    assert len(params) >= 40 and set(params.values()) <= {'rw', 'r', 'w'}
    header = open(os.path.join(ROOT, 'include', 'blueice_hip.h')).read()
    for name in params:
        assert re.search(r'\b%s\b' % re.escape(name), header), '%s is not documented in include/blueice_hip.h' % name
    small = C.create_string_buffer(8)                    # truncation keeps the terminator and still reports the size
    assert lib.bi_list_params(small, 8) == n and len(small.value) == 7


def test_library_override_is_not_a_backend_switch():
    """BLUEICE_AMD_LIB lets an A/B run load another build of libblueice_hip; a library that does not say it is the gfx950
    build -- the host build of the boundary tests, say -- is refused (a latent backend switch)."""
    import subprocess
    import sys
    from blueice_amd import build
    host = build.build_host()
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from blueice_amd import _capi\n"
            "from blueice_amd.exceptions import DeviceError\n"
            "try:\n    _capi.load()\nexcept DeviceError as e:\n    print('refused:', e)\n") % ROOT
    res = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, BLUEICE_AMD_LIB=host), capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    assert res.stdout.startswith('refused:'), res.stdout


def _compile_c_demo(tmp_path):
    import subprocess
    exe = str(tmp_path / 'c_abi_demo')
    libdir = os.path.join(ROOT, 'blueice_amd', 'lib')
    cmd = ['gcc', '-O2', '-Wall', '-Werror', '-std=c99', '-I' + os.path.join(ROOT, 'include'),
           os.path.join(ROOT, 'examples', 'c_abi_demo.c'), '-o', exe, '-L' + libdir, '-lblueice_hip', '-lm',
           '-Wl,-rpath,' + libdir]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    return exe


def test_header_is_plain_c_and_a_c_program_links(lib, tmp_path):
    """include/blueice_hip.h compiles as C99 and a plain C caller links against the library (no C++ / HIP / torch
    types in the boundary)."""
    _compile_c_demo(tmp_path)


@pytest.mark.gpu
def test_c_program_against_the_oracle(lib, tmp_path):
    """examples/c_abi_demo.c -- a C caller of the ABI, no Python in the loop -- reproduces the oracle on the model it
    builds (BASELINE.json configs[0] shape: 2 sources, 3 anchors, 40 bins), including the out-of-box point."""
    import subprocess
    import numpy as np
    from oracle import blueice_oracle as orc
    exe = _compile_c_demo(tmp_path)
    res = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    rows = np.array([[float(v) for v in line.split()] for line in res.stdout.strip().splitlines()])
    assert rows.shape == (6, 5)
    S, B, anchor_z = 2, 40, np.array([-1.0, 0.0, 1.0])
    x = (np.arange(B) + 0.5) / B * 10.0 - 5.0
    ps = np.empty((3, S, B))
    mus = np.empty((3, S))
    for a, za in enumerate(anchor_z):
        g = np.exp(-0.5 * (x - 0.8 * za) ** 2)
        ps[a, 0] = g / g.sum()          # (the C program sums in index order; agreement is to rounding, see rtol)
        ps[a, 1] = 1.0 / B
        mus[a] = (1000.0 * (1.0 + 0.05 * za), 500.0)
    counts = np.floor((0.75 * mus[1, 0] + 0.25 * mus[2, 0]) * (0.75 * ps[1, 0] + 0.25 * ps[2, 0]) + 500.0 / B + 0.5)
    model = dict(anchor_z=[anchor_z], ps=ps, mus=mus, n_model=None)
    for z, r0, r1, ll, st in rows:
        want = orc.loglikelihood(model, counts, [z], [r0, r1])
        if np.isfinite(want):
            assert abs(ll - want) <= 1e-10 * abs(want) and st == 0
        else:
            assert ll == want and int(st) == 1
