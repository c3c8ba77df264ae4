"""The host build of the C ABI's minimal entry points (blueice_amd/csrc/host_backend.cpp -> libblueice_host.so; SURVEY.md
section 7 step 3: "a pure-C++ CPU backend behind the same ABI, lets every API test run without a GPU").  It is a
restatement of the reference's loops in C++, independent of the numpy/scipy oracle: here it is held against
  (1) every golden fixture the real reference produced (bit for bit -- it follows the reference's operation order),
  (2) the oracle on random models, and
  (3) the plain C caller of examples/c_abi_demo.c, which therefore runs in the CPU suite too.
The package never loads this library: the last test checks that."""
import os
import subprocess

import numpy as np
import pytest

from golden_util import case_names, load_case, rate_scale_of, same
from host_lib import HostContext, load
from oracle import blueice_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BI_ST_BB = 4 | 8


@pytest.fixture(scope='module')
def lib():
    return load()


def _context(lib, c):
    ctx = HostContext(lib)
    ctx.upload_model(c['model']['anchor_z'], c['model']['ps'], c['model']['mus'], c['model']['n_model'], c['bb_source'],
                     c['allow_negative'])
    ctx.upload_counts(c['counts'])
    return ctx


@pytest.mark.parametrize('name', case_names())
def test_reference_goldens_bit_for_bit(lib, name):
    c = load_case(name)
    ctx = _context(lib, c)
    for j, ll_ref in enumerate(c['call_ll']):
        ll, st = ctx.eval(c['call_z'][j], rate_scale_of(c, j))
        asserts = ('call_asserts_%d' % j) in c['raw'].files
        assert bool(st[0] & BI_ST_BB) == asserts, (name, j, st[0])
        if not asserts:
            assert same(ll[0], ll_ref, rtol=0.0), (name, j, ll[0], ll_ref)
    f = c['raw']
    for key in f.files:                                     # full_output: adjusted (mus, ps) as the reference returned them
        if key.startswith('full_') and key.endswith('_mus'):
            j = int(key.split('_')[1])
            ll, mus, ps, st = ctx.eval_full(c['call_z'][j], rate_scale_of(c, j))
            np.testing.assert_array_equal(mus, f['full_%d_mus' % j])
            np.testing.assert_array_equal(ps.reshape(f['full_%d_ps' % j].shape), f['full_%d_ps' % j])
    ctx.close()


def test_random_models_against_the_oracle(lib):
    rng = np.random.default_rng(12)
    for trial in range(30):
        d = int(rng.integers(0, 4))
        S = int(rng.integers(1, 5))
        bins = tuple(int(b) for b in rng.integers(1, 40, size=int(rng.integers(1, 3))))
        grids = [np.sort(rng.uniform(-3, 3, size=int(rng.integers(1, 5)))) for _ in range(d)]
        shape = tuple(len(g) for g in grids)
        ps = rng.random(shape + (S,) + bins) + 0.01
        ps /= ps.reshape(shape + (S, -1)).sum(-1).reshape(shape + (S,) + (1,) * len(bins))
        mus = rng.uniform(5, 400, shape + (S,))
        bb = int(rng.integers(-1, S)) if trial % 2 else -1
        nm = rng.integers(1, 60, shape + (S,) + bins).astype(float) if bb >= 0 else None
        counts = rng.poisson(3.0, bins).astype(float)
        if trial % 5 == 0:
            counts.flat[0] = 0.5
        ctx = HostContext(lib)
        ctx.upload_model(grids, ps, mus, nm, bb)
        ctx.upload_counts(counts)
        model = dict(anchor_z=grids, ps=ps, mus=mus, n_model=nm)
        for _ in range(6):
            z = np.array([rng.uniform(g[0], g[-1]) if len(g) > 1 else g[0] for g in grids])
            if d and rng.random() < 0.2:
                z[0] = grids[0][-1] + (1.0 if rng.random() < 0.5 else 0.0)       # top edge / outside
            r = rng.uniform(0, 2, S)
            if rng.random() < 0.15:
                r[0] = -1.0
            ll, st = ctx.eval(z, r)
            try:
                want = orc.loglikelihood(model, counts, z, r, bb_source=bb if bb >= 0 else None)
            except AssertionError:                       # one of the reference's Beeston-Barlow asserts
                assert st[0] & BI_ST_BB, (trial, st[0])
                continue
            assert not (st[0] & BI_ST_BB), (trial, st[0])
            assert same(ll[0], want, rtol=0.0), (trial, ll[0], want)
        if d:
            z = np.array([0.5 * (g[0] + g[-1]) for g in grids])
            np.testing.assert_array_equal(ctx.interpolate(0, z), orc.interpolate(grids, ps, z))
            np.testing.assert_array_equal(ctx.interpolate(1, z), orc.interpolate(grids, mus, z))
            with pytest.raises(RuntimeError, match='out of bounds'):
                ctx.interpolate(0, z + 100.0)
        ctx.close()


def test_state_and_argument_errors(lib):
    import ctypes as C
    ctx = HostContext(lib)
    out = np.empty(1)
    assert lib.bi_eval(ctx.h, 1, None, None, None, out.ctypes.data_as(C.c_void_p), None) == -3        # BI_ERR_STATE: no model
    ctx.upload_model([np.array([0.0, 1.0])], np.full((2, 1, 3), 1 / 3), np.full((2, 1), 10.0))
    assert lib.bi_eval(ctx.h, 1, None, None, None, out.ctypes.data_as(C.c_void_p), None) == -3        # no data
    ctx.upload_counts(np.array([1.0, 2.0, 3.0]))
    ll, st = ctx.eval([0.25], dataset=[5])
    assert ll[0] == -np.inf and st[0] == 16
    ll, st = ctx.eval([np.nan])
    assert ll[0] == -np.inf and st[0] == 1
    assert b'host' in lib.bi_version() and b'not the product' in lib.bi_version()
    ctx.close()


def test_c_program_runs_on_the_host_build(lib, tmp_path):
    """examples/c_abi_demo.c -- a C99 caller of the ABI -- linked against the host build reproduces the oracle."""
    from blueice_amd import build
    libdir = os.path.dirname(build.build_host())
    exe = str(tmp_path / 'c_abi_demo_host')
    cmd = ['gcc', '-O2', '-Wall', '-Werror', '-std=c99', '-I' + os.path.join(ROOT, 'include'),
           os.path.join(ROOT, 'examples', 'c_abi_demo.c'), '-o', exe, '-L' + libdir, '-lblueice_host', '-lm', '-Wl,-rpath,' + libdir]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
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
        ps[a, 0] = g / g.sum()
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


def test_the_package_never_loads_the_host_build():
    """No CPU fallback: the product binds libblueice_hip.so only."""
    pkg = os.path.join(ROOT, 'blueice_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py') and fn != 'build.py':
            assert 'blueice_host' not in open(os.path.join(pkg, fn)).read(), fn
    assert 'blueice_host' not in open(os.path.join(ROOT, 'bench.py')).read()
