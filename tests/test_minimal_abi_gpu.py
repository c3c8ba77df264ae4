"""The minimal entry points of the C ABI -- exactly what the reference-side binding of INTEGRATION.md section B calls
(bi_upload_model / bi_model_*, bi_set_allow_negative, bi_upload_counts, bi_eval, bi_eval_full) -- on the PRODUCT library,
through the same raw ctypes binding that holds the host build against the reference's goldens in the CPU suite
(tests/host_lib.py; the blueice_amd package is not involved).  In the build container the reference's own tests run over
these entry points of the host build (tools/run_reference_tests_over_stub.py); this is the GPU half of that statement:
libblueice_hip.so, same calls, every golden fixture the real reference produced."""
import numpy as np
import pytest

from golden_util import case_names, load_case, rate_scale_of, same
from host_lib import HostContext, load

pytestmark = pytest.mark.gpu
BI_ST_BB = 4 | 8


@pytest.fixture(scope='module')
def lib():
    from blueice_amd._capi import LIB_PATH
    lib = load(LIB_PATH)
    assert b'not the product' not in lib.bi_version()
    return lib


@pytest.mark.parametrize('name', case_names())
def test_reference_goldens_through_the_minimal_entry_points(lib, name):
    c = load_case(name)
    ctx = HostContext(lib)
    ctx.upload_model(c['model']['anchor_z'], c['model']['ps'], c['model']['mus'], c['model']['n_model'], c['bb_source'],
                     c['allow_negative'])
    ctx.upload_counts(c['counts'])
    for j, ll_ref in enumerate(c['call_ll']):
        ll, st = ctx.eval(c['call_z'][j], rate_scale_of(c, j))
        asserts = ('call_asserts_%d' % j) in c['raw'].files
        assert bool(st[0] & BI_ST_BB) == asserts, (name, j, st[0])
        if not asserts:
            assert same(ll[0], ll_ref, rtol=1e-10), (name, j, ll[0], ll_ref)
    f = c['raw']
    for key in f.files:                                     # full_output: the interpolated (mus, ps) are scipy's bits
        if key.startswith('full_') and key.endswith('_mus'):
            j = int(key.split('_')[1])
            ll, mus, ps, st = ctx.eval_full(c['call_z'][j], rate_scale_of(c, j))
            np.testing.assert_array_equal(mus, f['full_%d_mus' % j])
            if c['bb_source'] >= 0:                         # (the Beeston-Barlow adjusted pmf: the kernel's roots, 1 ulp)
                np.testing.assert_allclose(ps.reshape(f['full_%d_ps' % j].shape), f['full_%d_ps' % j], rtol=1e-13, atol=0)
            else:
                np.testing.assert_array_equal(ps.reshape(f['full_%d_ps' % j].shape), f['full_%d_ps' % j])
    ctx.close()
