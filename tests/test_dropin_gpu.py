"""The drop-in on the GPU: blueice_amd.BinnedLogLikelihood built from the same configs as the
reference (tests/model_zoo.py) must return the reference's numbers (tests/golden) through the same
API -- lf(**kw), full_output, bestfit_scipy -- and its batched entry points must agree with the
scalar calls.  Tolerance 1e-10 * max(1, |ref|); +-inf / nan / raised errors exact."""
import numpy as np
import pytest
from scipy import stats

import model_zoo
from golden_util import GOLDEN_DIR, load_case, same

pytestmark = pytest.mark.gpu
RTOL = 1e-10


@pytest.fixture(scope='module')
def ns():
    return model_zoo.namespace_of('blueice_amd')


@pytest.mark.parametrize('name', list(model_zoo.CASES))
def test_lf_calls_match_reference(ns, name):
    lf, calls, full = model_zoo.CASES[name](ns)
    c = load_case(name)
    scalar = []
    for j, kw in enumerate(calls):
        if ('call_asserts_%d' % j) in c['raw'].files:
            with pytest.raises(AssertionError):
                lf(**kw)
            scalar.append(np.nan)
            continue
        ll = lf(**kw)
        assert same(ll, c['call_ll'][j], RTOL), (name, j, kw, ll, c['call_ll'][j])
        scalar.append(ll)
    for j in full:
        if ('full_%d_mus' % j) not in c['raw'].files:
            continue
        ll, mus, ps = lf(full_output=True, **calls[j])
        assert same(ll, c['call_ll'][j], RTOL)
        np.testing.assert_allclose(mus, c['raw']['full_%d_mus' % j], rtol=1e-12)
        assert ps.shape == c['raw']['full_%d_ps' % j].shape
        np.testing.assert_allclose(ps, c['raw']['full_%d_ps' % j], rtol=1e-12, atol=1e-300)
    # batched form of the same calls (no livetime / asserting calls in the batch)
    keep = [j for j, kw in enumerate(calls) if 'livetime_days' not in kw and not np.isnan(scalar[j])]
    names = sorted({k for j in keep for k in calls[j]})
    if keep and names:
        _, defaults = lf._kwargs_to_settings()
        pts = {}
        for n in names:
            dflt = defaults.get(n, 1.0)
            pts[n] = [calls[j].get(n, dflt) for j in keep]
        batch = lf.eval_points(pts)
        for b, j in zip(batch, keep):
            assert same(b, scalar[j], 1e-12), (name, j, b, scalar[j])


@pytest.mark.parametrize('name', list(model_zoo.API_CASES))
def test_api_cases_match_reference(ns, name):
    """Calls that are not functions of the anchor tensors alone -- compute_pdf (in and outside the anchor box, with
    priors), zero live time -- against what the reference returned or raised (tests/golden/api_*.npz)."""
    lf, calls = model_zoo.API_CASES[name](ns)
    f = np.load(GOLDEN_DIR + '/%s.npz' % name)
    assert len(calls) == len(f['call_ll'])
    errors = {'ValueError': ValueError, 'AssertionError': AssertionError, 'NotImplementedError': NotImplementedError}
    for j, kw in enumerate(calls):
        err = str(f['call_error'][j])
        if err:
            with pytest.raises(errors[err]):
                lf(**kw)
            continue
        ll = lf(**kw)
        assert same(ll, f['call_ll'][j], RTOL), (name, j, kw, ll, f['call_ll'][j])
    if name.startswith('api_livetime_zero'):
        pts = lf.eval_points(dict(shift=[0.3, -0.5, 1.5]), livetime_days=0.)
        want = 0.0 if name == 'api_livetime_zero' else -np.inf
        assert pts[0] == want and pts[1] == want and pts[2] == -np.inf
        with pytest.raises(ValueError):
            lf.eval_points(dict(shift=[0.3]), livetime_days=1.)


def test_source_wise_interpolation_as_the_reference_tests_it(ns):
    """tests/test_likelihood.py::test_source_wise_interpolation of the reference: when every source responds to every
    shape parameter, source-wise interpolation returns exactly what the plain likelihood returns (value, rates,
    per-event densities)."""
    data = np.zeros(5, dtype=[('x', float), ('source', int)])
    data['x'] = np.linspace(0, 1, 5)
    out = []
    for source_wise in (False, True):
        config = ns.conf_for_test(events_per_day=1)
        if source_wise:
            config['source_wise_interpolation'] = True
        lf = ns.UnbinnedLogLikelihood(config)
        lf.add_shape_parameter('mu', anchors={-2: -2, 0: 0, 2: 2})
        lf.prepare()
        lf.set_data(data)
        out.append((lf(full_output=True), lf(full_output=True, mu=1)))
    for plain, sw in zip(*out):
        assert plain[0] == sw[0]
        assert (plain[1] == sw[1]).all() and (plain[2] == sw[2]).all()
    # ... and the batched form agrees with single calls on the three-source case of the goldens
    lf, calls = model_zoo.api_source_wise(ns)
    keep = [kw for kw in calls if np.isfinite(lf(**kw))]
    names = ['mu', 'sigma', 'a_rate_multiplier', 'b_rate_multiplier', 'c_rate_multiplier']
    dflt = dict(mu=0., sigma=1.)
    pts = {n: [kw.get(n, dflt.get(n, 1.0)) for kw in keep] for n in names}
    for b, kw in zip(lf.eval_points(pts), keep):
        assert same(b, lf(**kw), 1e-12)


def test_reparam_as_the_reference_tests_it(ns):
    """tests/test_likelihood_reparam.py of the reference: closed-form values, agreement with the wrapped likelihood,
    the consistency assertions, and a fit in the new parameters."""
    from copy import deepcopy
    from scipy import stats
    from blueice_amd import LogLikelihoodReParam
    from blueice_amd.test_helpers import BASE_CONV_CONFIG, conf_for_reparam_test

    def fresh():
        lf_old = ns.UnbinnedLogLikelihood(conf_for_reparam_test(events_per_day=1))
        for n in ('op0', 'op1', 'op2'):
            lf_old.add_rate_parameter(n)
        lf_old.prepare()
        return lf_old

    lf_old = fresh()
    lf = LogLikelihoodReParam(lf_old, deepcopy(BASE_CONV_CONFIG))
    lf.set_data(np.zeros(3, dtype=[('x', float), ('source', int)]))
    for v in (1, 2, 3):
        total = v ** 2 + v ** 2 + v * v
        want = -total + 3 * np.log(total) + 3 * stats.norm.logpdf(0)
        assert np.isclose(lf(np0=v, np1=v), want, atol=1e-8)
    d = lf.base_model.simulate()
    lf.set_data(d)
    assert np.isclose(lf(), lf_old())
    assert np.isclose(lf(np0=2), lf_old(op0_rate_multiplier=4, op2_rate_multiplier=2))
    assert np.isclose(lf(np0=2, np1=2), lf_old(op0_rate_multiplier=4, op1_rate_multiplier=4, op2_rate_multiplier=4))
    assert list(lf.rate_parameters) == [] and list(lf.shape_parameters) == ['np0', 'np1']
    assert lf.get_bounds('np0') == (1e-12, 10) and len(lf.get_bounds()) == 2
    assert len(lf.base_model.simulate(dict(np0=3., np1=0.5))) >= 0           # simulate() speaks the new parameters
    bad = deepcopy(BASE_CONV_CONFIG)
    bad['op2_rate_multiplier'] = dict(params=['np0', 'np7'], func=lambda a, b: a * b)
    with pytest.raises(AssertionError):
        LogLikelihoodReParam(fresh(), bad)
    conf = conf_for_reparam_test(events_per_day=1)
    del conf['np1']
    inner = ns.UnbinnedLogLikelihood(conf)
    inner.prepare()
    with pytest.raises(AssertionError):
        LogLikelihoodReParam(inner, deepcopy(BASE_CONV_CONFIG))
    # the inference helpers are methods of the wrapper too
    np.random.seed(3)
    lf.set_data(lf.base_model.simulate(dict(np0=2., np1=1.5), livetime_days=40))
    fit, ll_max = lf.bestfit_scipy(pass_bounds_to_minimizer=True)
    assert set(fit) == {'np0', 'np1'} and ll_max >= lf(np0=2., np1=1.5) - 1e-9


def test_gradient_fit_falls_back_without_gradient_support(ns):
    """bestfit_scipy(use_gradient=True) on likelihoods that have no analytic gradient (a sum containing a term without one:
    here an ancillary Python function) must take the numerical route instead of failing mid-fit.  Beeston-Barlow likelihoods
    have one since round 3, unbinned likelihoods since round 4 -- and sums of those do too."""
    from blueice_amd import LogAncillaryLikelihood, LogLikelihoodSum
    bb, _, _ = model_zoo.bb_two_shape(ns)
    plain, _, _ = model_zoo.c1_like(ns)
    unb, _, _ = model_zoo.unb_shape_2src(ns)
    assert plain.supports_gradient and bb.supports_gradient and unb.supports_gradient
    fixed = dict(sigma=1.2, some_multiplier=0.8, shift=0.2)
    total = LogLikelihoodSum([plain, unb])
    assert total.supports_gradient and LogLikelihoodSum([plain, bb]).supports_gradient
    v, g = total.value_and_gradient(**fixed)
    assert same(v, total(**fixed), 1e-12) and all(np.isfinite(list(g.values())))
    a = total.bestfit_scipy(use_gradient=True, **fixed)
    b = total.bestfit_scipy(**fixed)
    assert abs(a[1] - b[1]) <= 1e-6 * abs(b[1]) or a[1] > b[1]
    anc = LogAncillaryLikelihood(lambda values: -0.5 * (values['shift'] / 0.3) ** 2, ['shift'], config=dict(shift=0.0))
    mixed = LogLikelihoodSum([plain, anc])
    assert not mixed.supports_gradient
    with pytest.raises(NotImplementedError):
        mixed.value_and_gradient()
    a = mixed.bestfit_scipy(use_gradient=True)
    b = mixed.bestfit_scipy()
    assert same(a[1], b[1], 1e-9)
    # a sum with a Beeston-Barlow term: the analytic route and the differencing route reach the same maximum
    both = LogLikelihoodSum([plain, bb])
    fixed = dict(strlen_multiplier=2, dummy=0.5, shift=0.1)
    a = both.bestfit_scipy(use_gradient=True, **fixed)
    b = both.bestfit_scipy(**fixed)
    assert abs(a[1] - b[1]) <= 1e-6 * abs(b[1])


def test_reference_binned_tests_verbatim(ns):
    """tests/test_binned_likelihood.py::test_single_bin / test_multi_bin of the reference,
    with the reference's own assertions (closed-form scipy expectations)."""
    from blueice_amd import BinnedLogLikelihood
    from blueice_amd.test_helpers import conf_for_test, make_data, almost_equal, FixedSampleSource
    lf = BinnedLogLikelihood(conf_for_test(mc=True, analysis_space=[['x', [-40, 40]]]))
    lf.add_rate_parameter('s0')
    lf.prepare()
    lf.set_data(np.zeros(1, dtype=[('x', float), ('source', int)]))
    assert almost_equal(lf(), stats.poisson(1000).logpmf(1), 1e-12)
    assert almost_equal(lf(s0_rate_multiplier=5.4), stats.poisson(5400).logpmf(1), 1e-12)
    lf.set_data(np.zeros(0, dtype=[('x', float), ('source', int)]))
    assert lf(s0_rate_multiplier=0.) == stats.poisson(0).logpmf(0) == 0         # test_zero_bin

    mc = [dict(n_events=24, x=0.5, y=0.5), dict(n_events=56, x=1.5, y=0.5),
          dict(n_events=6, x=0.5, y=2), dict(n_events=14, x=1.5, y=2)]
    data, n_mc = make_data(mc)
    conf = conf_for_test(events_per_day=42, default_source_class=FixedSampleSource, data=data,
                         analysis_space=[['x', [0, 1, 5]], ['y', [0, 1, 4]]])
    lf = BinnedLogLikelihood(conf)
    lf.add_rate_parameter('s0')
    lf.add_shape_parameter('strlen_multiplier', {1: 'x', 2: 'hi', 3: 'wha'}, base_value=1)
    lf.prepare()
    obs = [dict(n_events=18, x=0.5, y=0.5), dict(n_events=70, x=1.5, y=0.5),
           dict(n_events=4, x=0.5, y=2), dict(n_events=10, x=1.5, y=2)]
    lf.set_data(make_data(obs)[0])
    mus = [42 / n_mc * m['n_events'] for m in mc]
    seen = [o['n_events'] for o in obs]
    for z in (1, 2, 2.3):
        assert almost_equal(lf(strlen_multiplier=z),
                            np.sum([stats.poisson(z * mu).logpmf(k) for mu, k in zip(mus, seen)]))
    with pytest.raises(NotImplementedError):
        lf(compute_pdf=True, strlen_multiplier=2)
    assert lf() == lf(strlen_multiplier=1)                                       # base_value default


def test_error_handling_and_unphysical(ns):
    from blueice_amd.exceptions import InvalidParameter, NotPreparedException
    lf, calls, _ = model_zoo.c1_like(ns)
    with pytest.raises(InvalidParameter):
        lf(blargh=41)
    with pytest.raises(ValueError):
        lf(shift='hi')
    assert lf(s0_rate_multiplier=-0.1) == -np.inf
    lf.config['unphysical_behaviour'] = 'error'
    with pytest.raises(ValueError):
        lf(s0_rate_multiplier=-0.1)
    with pytest.raises(ValueError):
        lf.eval_points(dict(s0_rate_multiplier=[1., -0.1]))
    del lf.config['unphysical_behaviour']
    assert lf(shift=1.2) == -np.inf
    lf.prepare()                     # re-preparing invalidates the data (likelihood.py:253)
    with pytest.raises(NotPreparedException):
        lf()


def test_compute_pdf_numeric(ns):
    """compute_pdf=True builds the model at the requested point; at an anchor it must agree with the
    interpolated value (same templates), off-anchor it differs."""
    lf, _, _ = model_zoo.c1_like(ns)
    a = lf(shift=1.0, s0_rate_multiplier=1.3)
    b = lf(compute_pdf=True, shift=1.0, s0_rate_multiplier=1.3)
    assert same(b, a, 1e-12)
    assert abs(lf(compute_pdf=True, shift=0.5) - lf(shift=0.5)) > 1e-6


def test_priors_and_livetime(ns):
    lf, _, _ = model_zoo.d2_nonuniform(ns)
    base = lf(shift=0.3, stretch=3.3)
    lf.add_rate_uncertainty('s1', 0.2)
    lf.shape_parameters['shift'] = (lf.shape_parameters['shift'][0], stats.norm(0, 1).logpdf, None)
    got = lf(shift=0.3, stretch=3.3, s1_rate_multiplier=1.1)
    lf.rate_parameters['s1'] = None
    lf.shape_parameters['shift'] = (lf.shape_parameters['shift'][0], None, None)
    bare = lf(shift=0.3, stretch=3.3, s1_rate_multiplier=1.1)
    assert same(got, bare + stats.norm(1, 0.2).logpdf(1.1) + stats.norm(0, 1).logpdf(0.3), 1e-13)
    # livetime scaling == scaling every rate (single global factor), likelihood.py:374-382
    assert same(lf(shift=0.3, stretch=3.3, livetime_days=5.),
                lf(shift=0.3, stretch=3.3, s0_rate_multiplier=2.5, s1_rate_multiplier=2.5, s2_rate_multiplier=2.5), 1e-13)
    assert base == lf(shift=0.3, stretch=3.3)


def test_bestfit_scipy_matches_reference(ns):
    f = np.load(GOLDEN_DIR + '/fit_c1_like.npz')
    lf = model_zoo.fit_c1_like(ns)
    res, ll = lf.bestfit_scipy()
    assert list(res.keys()) == [str(x) for x in f['fit_all_names']]
    assert abs(ll - float(f['fit_all_ll'])) < 1e-6 * abs(ll)
    np.testing.assert_allclose(list(res.values()), f['fit_all_values'], rtol=2e-3, atol=2e-3)
    res, ll = lf.bestfit_scipy(shift=0.2)
    assert list(res.keys()) == [str(x) for x in f['fit_fixshift_names']]
    assert abs(ll - float(f['fit_fixshift_ll'])) < 1e-6 * abs(ll)
    # "Don't fit": everything fixed -> ({}, lf(...)) exactly (inference.py:150-151)
    res, ll = lf.bestfit_scipy(shift=0.2, s0_rate_multiplier=1., s1_rate_multiplier=0.9)
    assert res == {} and same(ll, float(f['fit_none_ll']), RTOL)
    objective, names, guess, bounds = lf.make_objective()
    assert names == ['s0_rate_multiplier', 's1_rate_multiplier', 'shift']
    assert list(guess) == [1, 1, 0] and bounds == [(0, None), (0, None), (-1.0, 1.0)]
    assert same(objective(guess), -lf(), 0)
    best = lf.best_anchor()
    assert set(best) == {'shift'} and best['shift'] in (-1.0, 0.0, 1.0)
    assert lf(**best) == max(lf(shift=z) for z in (-1., 0., 1.))


def test_morpher_api_contract():
    """The reference's tests/test_morphers.py::test_morpher_api, for the device-backed morpher."""
    from collections import OrderedDict
    from blueice_amd import pdf_morphers
    from blueice_amd.exceptions import NoShapeParameters
    for name, cls in pdf_morphers.MORPHERS.items():
        with pytest.raises(NoShapeParameters):
            cls(config={}, shape_parameters=OrderedDict())
        sp = OrderedDict([('bla', ({-1: -1, 0: 0, 1: 1}, None, None))])
        mr = cls(config={}, shape_parameters=sp)
        aps = mr.get_anchor_points(bounds=[(-1, 1)], n_models=3)
        assert isinstance(aps, list) and isinstance(aps[0], tuple)
        scalar_itp = mr.make_interpolator(lambda _: 0, extra_dims=[], anchor_models={z: None for z in aps})
        assert scalar_itp([0]) == 0
        matrix_itp = mr.make_interpolator(lambda _: 0, extra_dims=[2, 2], anchor_models={z: None for z in aps})
        np.testing.assert_array_equal(matrix_itp([0]), np.zeros((2, 2)))
        lin = mr.make_interpolator(lambda m: np.array([m, 2. * m]), extra_dims=[2],
                                   anchor_models={z: float(z[0]) for z in aps})
        np.testing.assert_allclose(lin([0.25]), [0.25, 0.5], rtol=1e-15)
        with pytest.raises(ValueError):
            lin([1.5])


def test_toys_match_scalar_calls(ns):
    lf, _, _ = model_zoo.d3_small(ns)
    rng = np.random.default_rng(5)
    base = lf.data_events_per_bin.histogram
    toys = rng.poisson(base + 0.3, size=(7,) + base.shape).astype(float)
    want = []
    for t in toys:
        lf.set_binned_data(t)
        want.append(lf(shift=0.2, stretch=-0.4, tilt=0.9, s2_rate_multiplier=1.3))
    lf.set_binned_data(toys)
    got = lf.eval_toys(shift=0.2, stretch=-0.4, tilt=0.9, s2_rate_multiplier=1.3)
    for g, w in zip(got, want):
        assert same(g, w, 1e-12)
    assert same(lf(shift=0.2, stretch=-0.4, tilt=0.9, s2_rate_multiplier=1.3), want[0], 0)   # dataset 0
    both = lf.eval_points(dict(shift=[0.2, 0.2], stretch=-0.4, tilt=0.9, s2_rate_multiplier=1.3), dataset=[3, 6])
    assert same(both[0], want[3], 1e-12) and same(both[1], want[6], 1e-12)


def test_intervals_and_scans_match_reference(ns):
    """one_parameter_interval (blueice/inference.py:332-389) against the reference's own numbers, and the
    batched likelihood-ratio scan against scalar calls."""
    f = np.load(GOLDEN_DIR + '/fit_c1_like.npz')
    lf = model_zoo.fit_c1_like(ns)
    up = lf.one_parameter_interval('s1_rate_multiplier', bound=50., kind='upper', confidence_level=0.9)
    assert abs(up - float(f['upper_s1_90'])) < 2e-3 * float(f['upper_s1_90'])
    lo, hi = lf.one_parameter_interval('s0_rate_multiplier', bound=(2., 20.), kind='central', confidence_level=0.68,
                                       s1_rate_multiplier=0.)
    np.testing.assert_allclose([lo, hi], f['central_s0_68'], rtol=2e-3)
    # options of the fit routine travel through the driver's kwargs, as in the reference
    up2 = lf.one_parameter_interval('s1_rate_multiplier', bound=50., kind='upper', confidence_level=0.9, pass_bounds_to_minimizer=True)
    assert abs(up2 - float(f['upper_s1_90'])) < 5e-3 * float(f['upper_s1_90'])
    # nothing left to fit -> the whole grid in one device call
    s0 = np.linspace(5., 10., 11)
    sh = np.linspace(-0.5, 0.9, 8)
    grid = lf.likelihood_ratio_scan(('s0_rate_multiplier', s0), ('shift', sh), s1_rate_multiplier=0.)
    assert grid.shape == (11, 8) and grid.min() == 0
    ll = np.array([[lf(s0_rate_multiplier=a, shift=b, s1_rate_multiplier=0.) for b in sh] for a in s0])
    np.testing.assert_allclose(grid, ll.max() - ll, rtol=1e-10, atol=1e-9)
    # one parameter profiled out at every grid point
    prof = lf.likelihood_ratio_scan(('shift', sh[:4]), s1_rate_multiplier=0.)
    want = np.array([lf.bestfit_scipy(shift=b, s1_rate_multiplier=0.)[1] for b in sh[:4]])
    np.testing.assert_allclose(prof, want.max() - want, atol=1e-6)


def test_synthetic_model_through_the_plugin_api():
    """SyntheticModel.likelihood(): the synthetic tensors behind Source / Model / BinnedLogLikelihood (the route a
    user's sources take), against the oracle on the dense tensors and against a fit started elsewhere."""
    from oracle import blueice_oracle as orc
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('mini3')
    lf = m.likelihood()
    counts = m.counts(dense=True)
    lf.set_binned_data(counts.reshape(m.bins))
    model = m.dense_model()
    rng = np.random.default_rng(3)
    for _ in range(5):
        z = rng.uniform(-1, 1, m.d)
        r = rng.uniform(0.5, 1.5, m.S)
        kw = {'shape%d' % i: z[i] for i in range(m.d)}
        kw.update({'s%d_rate_multiplier' % s: r[s] for s in range(m.S)})
        want = orc.loglikelihood(model, counts, z, r)
        got = lf(**kw)
        assert abs(got - want) <= 1e-10 * abs(want)
    assert lf(shape0=1.5) == -np.inf                      # outside the anchor box (mini3 has anchors -1, 0, 1)
    # the four synthetic sources are the same noise, so their rates are degenerate: float one rate and one shape
    fixed = dict(s1_rate_multiplier=1, s2_rate_multiplier=1, s3_rate_multiplier=1, shape1=0.2, shape2=-0.3)
    best, ll = lf.bestfit_scipy(**fixed)
    best_g, ll_g = lf.bestfit_scipy(use_gradient=True, **fixed)
    assert set(best) == {'s0_rate_multiplier', 'shape0'}
    assert abs(ll - ll_g) <= 1e-6 * abs(ll)
    assert all(abs(best[k] - best_g[k]) < 5e-3 for k in best)
    assert ll >= lf(**fixed)
    z = np.array([best['shape0'], 0.2, -0.3])
    r = np.array([best['s0_rate_multiplier'], 1, 1, 1])
    assert abs(ll - orc.loglikelihood(model, counts, z, r)) <= 1e-10 * abs(ll)


def test_eval_in_two_halves_and_overlapped_sum():
    """bi_eval_begin / bi_eval_end: the same numbers as bi_eval, parked answers for rejected points, state errors;
    a LogLikelihoodSum of device likelihoods (one context per term) launches all terms before collecting any."""
    from blueice_amd import LogLikelihoodSum
    from blueice_amd.device import DeviceContext
    from blueice_amd.exceptions import NotPreparedException as StateError      # what BI_ERR_STATE raises
    from blueice_amd.synthetic import SyntheticModel
    m = SyntheticModel.named('mini3')
    ctx = DeviceContext(0)
    m.upload(ctx)
    ctx.upload_counts(m.counts(dense=True))
    z, r = m.random_points(20, seed=4)
    for sparse in (0, 1):
        ctx.set_param('sparse', sparse)
        for i in range(len(z)):
            want, st = ctx.eval(z[i], r[i])
            ctx.eval_begin(z[i], r[i])
            got, st2 = ctx.eval_end()
            assert got == want[0] and st2 == st[0]
    ctx.eval_begin(np.array([9.0, 0.0, 0.0]), r[0])              # outside the anchor box: nothing is launched
    assert ctx.eval_end() == (-np.inf, 1)
    ctx.eval_begin(z[0], -r[0])                                   # unphysical rates
    assert ctx.eval_end() == (-np.inf, 2)
    with pytest.raises(StateError):
        ctx.eval_end()                                            # nothing outstanding
    ctx.eval_begin(z[0], r[0])
    with pytest.raises(StateError):
        ctx.eval_begin(z[1], r[1])                                # one at a time
    with pytest.raises(StateError):
        ctx.eval(z[1], r[1])                                      # nor anything else in between
    ll, _ = ctx.eval_end()
    assert ll == ctx.eval(z[0], r[0])[0][0]
    ctx.close()

    terms = [SyntheticModel.named('mini3', seed=s).likelihood() for s in (1, 2, 3)]
    for s, lf in zip((1, 2, 3), terms):
        lf.set_binned_data(SyntheticModel.named('mini3', seed=s).counts(dense=True).reshape(m.bins))
    tot = LogLikelihoodSum(terms, likelihood_weights=[1, 0.5, 2])
    for kw in (dict(), dict(shape0=0.3, s1_rate_multiplier=1.2), dict(shape1=-0.7, shape2=0.9, s0_rate_multiplier=0.0)):
        want = sum(w * lf(**kw) for w, lf in zip([1, 0.5, 2], terms))
        assert tot(**kw) == want
    assert tot(shape0=5.0) == -np.inf                             # one call outside the box: every term parks -inf
    with pytest.raises(ValueError):                               # raised by the second half of the terms' host work
        tot(shape0='not a number')
    assert tot() == sum(w * lf() for w, lf in zip([1, 0.5, 2], terms))     # ... and nothing was left outstanding


def test_sum_of_likelihoods_batched_and_gradient_forms():
    """LogLikelihoodSum.eval_points / value_and_gradient = the weighted sums of the terms'; a gradient fit of the
    sum lands on the plain fit's maximum."""
    from blueice_amd import LogLikelihoodSum
    from blueice_amd.synthetic import SyntheticModel
    terms, weights = [], [1, 0.5]
    for s in (1, 2):
        m = SyntheticModel.named('mini3', seed=s)
        lf = m.likelihood()
        lf.set_binned_data(m.counts(dense=True).reshape(m.bins))
        terms.append(lf)
    tot = LogLikelihoodSum(terms, likelihood_weights=weights)
    rng = np.random.default_rng(2)
    pts = dict(shape0=rng.uniform(-1, 1, 50), s0_rate_multiplier=rng.uniform(0.5, 1.5, 50), shape2=0.25)
    got = tot.eval_points(pts)
    want = np.array([tot(shape0=a, s0_rate_multiplier=b, shape2=0.25) for a, b in zip(pts['shape0'], pts['s0_rate_multiplier'])])
    np.testing.assert_allclose(got, want, rtol=1e-13)
    kw = dict(shape0=0.3, shape1=-0.2, s0_rate_multiplier=1.1)
    v, g = tot.value_and_gradient(**kw)
    assert abs(v - tot(**kw)) <= 1e-12 * abs(v)
    for name in ('shape0', 's0_rate_multiplier', 's2_rate_multiplier'):
        h = 1e-6
        up = tot(**dict(kw, **{name: kw.get(name, 1.0) + h}))
        dn = tot(**dict(kw, **{name: kw.get(name, 1.0) - h}))
        assert abs(g[name] - (up - dn) / (2 * h)) <= 1e-5 * max(1.0, abs(g[name]))
    fixed = dict(s1_rate_multiplier=1, s2_rate_multiplier=1, s3_rate_multiplier=1, shape1=0.2, shape2=-0.3)
    best, ll = tot.bestfit_scipy(**fixed)
    best_g, ll_g = tot.bestfit_scipy(use_gradient=True, **fixed)
    assert abs(ll - ll_g) <= 1e-6 * abs(ll) and all(abs(best[k] - best_g[k]) < 5e-3 for k in best)
