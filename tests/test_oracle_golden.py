"""Pin the CPU oracle (oracle/blueice_oracle.py) against
  (1) golden vectors produced by the real reference (tests/golden/*.npz),
  (2) scipy itself (the third-party home of the arithmetic), bit for bit,
  (3) the closed-form expectations of the reference's own tests."""
import numpy as np
import pytest
from scipy import stats
from scipy.interpolate import RegularGridInterpolator

from oracle import blueice_oracle as orc
from golden_util import case_names, load_case, rate_scale_of, same, unbinned_case_names


@pytest.mark.parametrize('name', case_names())
def test_oracle_matches_reference_goldens(name):
    c = load_case(name)
    assert len(c['call_ll']) > 0
    for j, ll_ref in enumerate(c['call_ll']):
        bb = c['bb_source'] if c['bb_source'] >= 0 else None
        asserts = ('call_asserts_%d' % j) in c['raw'].files
        try:
            ll = orc.loglikelihood(c['model'], c['counts'], c['call_z'][j], rate_scale_of(c, j), bb_source=bb,
                                   allow_negative=c['allow_negative'])
        except AssertionError:
            assert asserts, "oracle asserted where the reference did not (call %d)" % j
            continue
        assert not asserts, "reference asserted where the oracle did not (call %d)" % j
        # the restatement is expected to be bit-identical; allow 1 ulp-scale slack only
        assert same(ll, ll_ref, rtol=4e-16), (name, j, ll, ll_ref)


@pytest.mark.parametrize('name', case_names())
def test_oracle_full_output(name):
    c = load_case(name)
    f = c['raw']
    for key in f.files:
        if not (key.startswith('full_') and key.endswith('_mus')):
            continue
        j = int(key.split('_')[1])
        z, rs = c['call_z'][j], rate_scale_of(c, j)
        mus = orc.rates_at(c['model'], z, rs)
        ps = orc.interpolate(c['model']['anchor_z'], c['model']['ps'], z)
        if c['bb_source'] >= 0:
            nm = orc.interpolate(c['model']['anchor_z'], c['model']['n_model'], z)
            mus, ps = orc.adjust_expectations_bb(mus, ps, nm, c['counts'], c['bb_source'])
        np.testing.assert_array_equal(mus, f['full_%d_mus' % j])
        np.testing.assert_array_equal(ps, f['full_%d_ps' % j])


def test_interpolate_bit_identical_to_scipy():
    rng = np.random.default_rng(0)
    for d, extra in [(1, (3,)), (2, (2, 5)), (3, (4, 3, 2)), (4, (2, 3)), (3, ())]:
        grids = [np.sort(rng.uniform(-3, 3, size=rng.integers(2, 6))) for _ in range(d)]
        vals = rng.normal(size=tuple(len(g) for g in grids) + extra)
        rgi = RegularGridInterpolator(grids, vals)
        pts = [np.array([rng.uniform(g[0], g[-1]) for g in grids]) for _ in range(40)]
        pts += [np.array([g[0] for g in grids]), np.array([g[-1] for g in grids]),
                np.array([g[len(g) // 2] for g in grids])]
        for z in pts:
            if d == 2 and extra == ():
                continue  # scipy uses a separate cython fast path for 2-d scalar fields
            np.testing.assert_array_equal(orc.interpolate(grids, vals, z), rgi(z)[0])
    with pytest.raises(ValueError):
        orc.interpolate([np.array([0., 1.])], np.zeros((2, 3)), [1.5])


def test_poisson_logpmf_bit_identical_to_scipy():
    rng = np.random.default_rng(1)
    mu = np.concatenate([rng.uniform(0, 50, 200), [0., 0., 0., -1., np.nan, np.inf, 1e-300, 5.]])
    k = np.concatenate([rng.poisson(20, 200).astype(float), [0., 3., 0.5, 2., 1., 1., 0., -1.]])
    with np.errstate(all='ignore'):
        ref = stats.poisson(mu).logpmf(k)
    got = orc.poisson_logpmf(k, mu)
    np.testing.assert_array_equal(got, ref)
    assert got[200] == 0 and got[201] == -np.inf and got[202] == -np.inf
    assert np.isnan(got[203]) and np.isnan(got[204])


def test_reference_test_expectations():
    """Closed forms asserted by the reference's tests/test_binned_likelihood.py and
    tests/test_BeestonBarlow.py, evaluated through the oracle on the golden tensors."""
    c = load_case('ref_single_bin')                       # test_single_bin: exact equality
    ll = [orc.loglikelihood(c['model'], c['counts'], c['call_z'][j], rate_scale_of(c, j)) for j in range(3)]
    assert ll[0] == stats.poisson(1000).logpmf(1)
    assert ll[1] == stats.poisson(5400).logpmf(1)
    c = load_case('ref_zero_bin')                         # test_zero_bin
    assert orc.loglikelihood(c['model'], c['counts'], c['call_z'][0], rate_scale_of(c, 0)) == 0.0
    c = load_case('ref_multi_bin')                        # test_multi_bin: anchors and off-anchor 2.3
    mus = np.array([42 / 100 * n for n in (24, 6, 56, 14)])    # bins in C order (x, y)
    seen = np.array([18, 4, 70, 10])
    for j, z in [(0, 1.), (1, 2.), (2, 2.3)]:
        want = np.sum(stats.poisson(z * mus).logpmf(seen))
        got = orc.loglikelihood(c['model'], c['counts'], c['call_z'][j], rate_scale_of(c, j))
        assert abs((got - want) / want) < 1e-6
    # Beeston-Barlow
    assert abs(orc.beeston_barlow_root2(np.array([32]), 0.2, np.array([1]), np.array([2]))[0] / 28.0814209 - 1) < 1e-6
    c = load_case('ref_bb_single_bin')
    A = (2 + 32) / (1 + 0.2)
    got = orc.loglikelihood(c['model'], c['counts'], c['call_z'][0], rate_scale_of(c, 0), bb_source=0)
    assert abs(got - stats.poisson(0.2 * A).logpmf(2)) < 1e-6 * abs(got)
    c = load_case('ref_bb_second_source')
    A_BB = orc.beeston_barlow_root2(np.array([16, 30, 32, 27]), 0.2, np.array([5, 7, 1, 3]), np.array([3, 5, 2, 7]))
    np.testing.assert_almost_equal(A_BB, [14.24, 26.8070, 28.08, 26.21], decimal=2)
    want = np.sum(stats.poisson(0.2 * A_BB + np.array([5, 7, 1, 3])).logpmf([3, 5, 2, 7]))
    got = orc.loglikelihood(c['model'], c['counts'], c['call_z'][0], rate_scale_of(c, 0), bb_source=0)
    assert abs((got - want) / want) < 1e-6
    assert got == -13.567903064008467                    # value traced in SURVEY.md section 8c
    c = load_case('bb_two_shape')
    assert c['call_ll'][1] == -22.706195309848717 and c['call_ll'][2] == -33.38765190325482


def test_edge_semantics():
    c = load_case('edge_mu_zero_hit')
    assert c['call_ll'][0] == -np.inf
    c = load_case('d3_small')
    assert c['call_ll'][-1] == -np.inf and c['call_ll'][-2] == -np.inf     # nan z, out-of-box z
    c = load_case('c1_like')
    assert c['call_ll'][8] == -np.inf                                      # negative rate


@pytest.mark.parametrize('name', unbinned_case_names())
def test_unbinned_oracle_matches_reference_goldens(name):
    """extended_loglikelihood path: oracle vs the reference's UnbinnedLogLikelihood."""
    c = load_case(name)
    assert c['kind'] == 1
    for j, ll_ref in enumerate(c['call_ll']):
        ll = orc.loglikelihood_unbinned(c['model'], c['call_z'][j], rate_scale_of(c, j), c['outlier'])
        assert same(ll, ll_ref, rtol=4e-16), (name, j, ll, ll_ref)
    for key in c['raw'].files:
        if key.startswith('full_') and key.endswith('_ps'):
            j = int(key.split('_')[1])
            ps = orc.interpolate(c['model']['anchor_z'], c['model']['ps'], c['call_z'][j])
            np.testing.assert_array_equal(ps, c['raw'][key])
    if name == 'unb_ref_value':            # the reference's own closed forms (test_likelihood_value)
        assert c['call_ll'][0] == -1 + stats.norm.logpdf(0)
        assert c['call_ll'][1] == -2 + np.log(2 * stats.norm.pdf(0))


def _pairwise(a, lo, n):
    """numpy's pairwise_sum (loops_utils.h.src) restated: what k_bb_chunk_sums / pairwise_sum_host reproduce."""
    if n < 8:
        res = 0.0
        for i in range(n):
            res += a[lo + i]
        return res
    if n <= 128:
        r = a[lo:lo + 8].copy()
        m = n - (n % 8)
        for row in a[lo + 8:lo + m].reshape(-1, 8):
            r = r + row
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        for i in range(m, n):
            res += a[lo + i]
        return res
    n2 = n // 2
    n2 -= n2 % 8
    return _pairwise(a, lo, n2) + _pairwise(a, lo + n2, n - n2)


def test_numpy_sum_order_the_device_reproduces():
    """The Beeston-Barlow normalisation N = n_model_events[i].sum() (blueice/likelihood.py:645) decides assertion
    outcomes by its last bit, so the device sums in NUMPY'S order: sequentially over chunks of 8192 elements, each chunk
    by pairwise_sum.  This pins that order for the numpy in use (if a numpy release changes it, this test says so before
    tests/test_bb_exact_gpu.py does)."""
    rng = np.random.default_rng(5)
    for n in (1, 7, 8, 9, 100, 128, 129, 1000, 8191, 8192, 8193, 20000, 65536, 250000):
        for shape in ((n,), (2, n)):
            a = rng.random(shape) * 50 + 1
            row = a if a.ndim == 1 else a[1]
            total = None
            for s in range(0, n, 8192):
                v = _pairwise(row, s, min(8192, n - s))
                total = v if total is None else total + v
            assert np.float64(total) == row.sum(), (n, shape)
    for shape in ((3, 7, 11, 13), (2, 37, 41, 29)):    # N-d rows, as n_model_events[i] is: summed as one contiguous run
        x = rng.random(shape) * 30 + 1
        flat = x[-1].ravel()
        total = None
        for s in range(0, flat.size, 8192):
            v = _pairwise(flat, s, min(8192, flat.size - s))
            total = v if total is None else total + v
        assert np.float64(total) == x[-1].sum(), shape
