"""Next row f-4 of SURVEY.md section 8: the extended unbinned likelihood and the likelihood sum on the device
path, against golden vectors from the reference's UnbinnedLogLikelihood and its own tests' closed forms."""
import numpy as np
import pytest
from scipy import stats

import model_zoo
from golden_util import load_case, rate_scale_of, same, unbinned_case_names

pytestmark = pytest.mark.gpu
RTOL = 1e-10


@pytest.fixture(scope='module')
def ns():
    return model_zoo.namespace_of('blueice_amd')


@pytest.mark.parametrize('name', unbinned_case_names())
def test_c_abi_matches_reference_goldens(name):
    from blueice_amd.device import DeviceContext
    c = load_case(name)
    ctx = DeviceContext(0)
    n_ev = c['bins'][0]
    grid_shape = tuple(len(g) for g in c['model']['anchor_z'])
    ctx.begin_model(c['model']['anchor_z'], c['S'], n_ev)
    ps = c['model']['ps'].reshape((-1, c['S'], n_ev)) if n_ev else np.zeros((int(np.prod(grid_shape)), c['S'], 0))
    mus = c['model']['mus'].reshape((-1, c['S']))
    for a in range(len(mus)):
        ctx.set_anchor(a, ps[a], mus[a])
    ctx.end_model()
    ctx.set_unbinned(c['outlier'])
    n = len(c['call_ll'])
    rs = np.array([rate_scale_of(c, j) for j in range(n)])
    batch, _ = ctx.eval(c['call_z'] if c['d'] else None, rs)
    for j in range(n):
        one, _ = ctx.eval(c['call_z'][j] if c['d'] else None, rs[j])
        assert same(one[0], c['call_ll'][j], RTOL), (name, j, one[0], c['call_ll'][j])
        assert same(batch[j], c['call_ll'][j], RTOL)
    with pytest.raises(ValueError):
        ctx.eval_datasets(c['call_z'][0] if c['d'] else None, rs[0])
    ctx.close()


@pytest.mark.parametrize('name', list(model_zoo.UNBINNED_CASES))
def test_unbinned_likelihood_class_matches_reference(ns, name):
    lf, calls, full = model_zoo.UNBINNED_CASES[name](ns)
    c = load_case(name)
    for j, kw in enumerate(calls):
        assert same(lf(**kw), c['call_ll'][j], RTOL), (name, j, kw)
    for j in full:
        ll, mus, ps = lf(full_output=True, **calls[j])
        assert same(ll, c['call_ll'][j], RTOL)
        np.testing.assert_allclose(mus, c['raw']['full_%d_mus' % j], rtol=1e-12)
        np.testing.assert_array_equal(ps, c['raw']['full_%d_ps' % j])


@pytest.mark.parametrize('dims,method,shapes,n_events', [(1, 'linear', 1, 200), (1, 'piecewise', 2, 200), (2, 'linear', 2, 200),
                                                         (3, 'linear', 1, 200), (3, 'piecewise', 0, 200), (2, 'linear', 0, 200),
                                                         (3, 'piecewise', 1, 6000), (3, 'linear', 1, 5000), (1, 'linear', 2, 4096)])
def test_events_scored_on_the_device_equal_host_scoring(ns, dims, method, shapes, n_events):
    """set_data of the unbinned likelihood for histogram-pdf sources (bi_score_events) against the host route
    (Model.score_events anchor by anchor, HistogramPdfSource.pdf = blueice/source.py:218-243): the same
    [anchor][source][event] tensor and the same likelihood.  Events on bin centres / edges / range limits included.
    Two-dimensional linear interpolation: scipy evaluates 2-D scalar fields with a separate routine that associates the
    products differently, so there the agreement is to rounding, elsewhere to the bit.  From 4096 events on the device
    orders the events by histogram cell before it gathers (round 4): per-event values still come back in the caller's order,
    to the bit; the likelihood, a sum over the events in another order, to 1e-12."""
    from collections import OrderedDict
    space = [['x', np.linspace(-4, 4, 17)], ['y', np.array([0., 0.4, 1., 2.2, 3.5, 5.])], ['w', np.linspace(-1, 1, 6)]][:dims]
    anchors = OrderedDict(list(OrderedDict(shift=(-1., 0., 1.), stretch=(0., 0.5, 1.)).items())[:shapes])
    out = []
    for on_device in (True, False):
        rng = np.random.default_rng(90 + dims)
        lf = model_zoo.morph_lf(ns, rng, 3, space, anchors, 4000, 150, unbinned=True,
                                lc=dict(device_scoring=on_device), extra_config=dict(pdf_interpolation_method=method))
        d = model_zoo.sample(rng, n_events, space)
        for nm, e in space:                                  # special places: limits, an inner edge, a bin centre
            d[nm][:4] = [e[0], e[-1], e[2], 0.5 * (e[1] + e[2])]
        lf.set_data(d)
        assert (lf._templates not in (None, False)) == on_device
        assert lf.ctx.get_param('events_sorted') == (1 if on_device and n_events >= 4096 else 0)
        calls = [{}, dict(s0_rate_multiplier=1.4, s2_rate_multiplier=0.3)]
        if shapes:
            calls += [dict(shift=0.35), dict(shift=-1., s1_rate_multiplier=0.)]
        if shapes > 1:
            calls += [dict(shift=0.6, stretch=0.8)]
        z = np.array([0.35, 0.8][:shapes])
        out.append(([lf(**kw) for kw in calls], lf.ps_interpolator(z) if shapes else lf(full_output=True)[2]))
        lf.set_data(d[:37])                                  # a second dataset re-uses the templates on the device
        out[-1] += (lf(),)
    (ll_dev, ps_dev, ll2_dev), (ll_host, ps_host, ll2_host) = out
    assert ps_dev.shape == ps_host.shape and np.all(np.isfinite(ps_dev)) and ps_dev.max() > 0
    if dims == 2 and method == 'linear':
        np.testing.assert_allclose(ps_dev, ps_host, rtol=1e-14, atol=0)
    else:
        np.testing.assert_array_equal(ps_dev, ps_host)
    for a, b in zip(ll_dev + [ll2_dev], ll_host + [ll2_host]):
        assert same(a, b, 1e-13 if n_events < 4096 else 1e-12), (a, b)


def test_device_scoring_is_skipped_where_it_does_not_apply(ns):
    """Analytic pdfs, overridden pdf() and non-finite event coordinates take the host route."""
    lf = ns.UnbinnedLogLikelihood(ns.conf_for_test(events_per_day=3.))          # GaussianSource: analytic pdf
    lf.add_shape_parameter('some_multiplier', (0.5, 1, 2))
    lf.prepare()
    lf.set_data(model_zoo._events([0.1, -0.4, 2.]))
    assert lf._templates is False and np.isfinite(lf())
    conf = ns.conf_for_test(mc=True, n_events_for_pdf=20000, events_per_day=3.)  # histogram pdf: device
    lf = ns.UnbinnedLogLikelihood(conf)
    lf.add_shape_parameter('mu', (-1., 0., 1.))
    lf.prepare()
    lf.set_data(model_zoo._events([0.1, -0.4, 2.]))
    assert lf._templates not in (None, False)
    value = lf(mu=0.3)
    lf.set_data(model_zoo._events([]))                                           # no events at all: ll = -sum of the rates
    assert lf._templates not in (None, False)
    assert same(lf(mu=0.3), -float(np.sum(lf.mus_interpolator(np.array([0.3])))), 1e-14) and np.isfinite(value)
    lf.prepare()                                                                 # a new prepare() drops the templates
    assert lf._templates is None


def test_reference_unbinned_tests_closed_forms(ns):
    """test_likelihood_value, test_rate_uncertainty, test_shape_uncertainty, test_multisource_likelihood,
    test_livetime_scaling, test_error_handling of the reference's tests/test_likelihood.py, restated."""
    from blueice_amd import UnbinnedLogLikelihood
    from blueice_amd.exceptions import InvalidParameter, InvalidParameterSpecification, NotPreparedException
    from blueice_amd.test_helpers import conf_for_test, almost_equal
    one = np.zeros(1, dtype=[('x', float), ('source', int)])
    lf = UnbinnedLogLikelihood(conf_for_test(events_per_day=1))
    lf.add_rate_uncertainty('s0', 0.5)
    lf.set_data(one)
    lp = stats.norm(1, 0.5).logpdf
    assert almost_equal(lf(), -1 + stats.norm.logpdf(0) + lp(1), 1e-13)
    assert almost_equal(lf(s0_rate_multiplier=2), -2 + np.log(2 * stats.norm.pdf(0)) + lp(2), 1e-13)

    lf = UnbinnedLogLikelihood(conf_for_test(events_per_day=1))
    with pytest.raises(InvalidParameterSpecification):
        lf.add_shape_uncertainty('strlen_multiplier', 0.5, {1: 'x', 2: 'hi', 3: 'wha'})
    lf.add_shape_uncertainty(setting_name='strlen_multiplier', fractional_uncertainty=0.5,
                             anchor_zs={1: 'x', 2: 'hi', 3: 'wha'}, base_value=1)
    lf.prepare()
    lf.set_data(one)
    assert almost_equal(lf(), -1 + stats.norm.logpdf(0) + lp(1), 1e-13)
    assert almost_equal(lf(strlen_multiplier=2), -2 + np.log(2 * stats.norm.pdf(0)) + lp(2), 1e-13)
    with pytest.raises(ValueError):
        lf(strlen_multiplier='hi')
    assert lf(strlen_multiplier=1.5) < lf()

    np.random.seed(4)
    lf = UnbinnedLogLikelihood(conf_for_test(n_sources=2))
    lf.add_shape_parameter('some_multiplier', (0.5, 1, 2, 4))
    lf.add_rate_parameter('s0')
    lf.add_rate_parameter('s1')
    with pytest.raises(NotPreparedException):
        lf.set_data(one)
    lf.prepare()
    with pytest.raises(NotPreparedException):
        lf()
    lf.set_data(lf.base_model.simulate())
    assert lf(s0_rate_multiplier=1, s1_rate_multiplier=1, some_multiplier=1) == lf()
    assert almost_equal(lf(s0_rate_multiplier=2), lf(s1_rate_multiplier=2), 1e-12)
    assert almost_equal(lf(s0_rate_multiplier=4), lf(s0_rate_multiplier=2.5, s1_rate_multiplier=2.5), 1e-12)
    assert almost_equal(lf(s0_rate_multiplier=2, s1_rate_multiplier=2), lf(some_multiplier=2), 1e-12)
    assert lf(some_multiplier=2) < lf()
    with pytest.raises(InvalidParameter):
        lf(blargh=41)

    conf = conf_for_test()
    lf0 = UnbinnedLogLikelihood(conf)
    lf0.prepare()
    lf0.set_data(one)
    with pytest.raises(ValueError):
        lf0(livetime_days=1)
    conf['livetime_days'] = 1
    lf = UnbinnedLogLikelihood(conf)
    lf.add_rate_parameter('s0')
    lf.prepare()
    lf.set_data(one)
    assert lf(livetime_days=1) == lf0()
    assert lf(livetime_days=2) == lf(s0_rate_multiplier=2)
    assert lf(livetime_days=0) == lf(s0_rate_multiplier=0)

    # compute_pdf=True builds the model at the requested point (test_noninterpolated_pdf)
    conf = conf_for_test(n_sources=1)
    conf['some_multiplier'] = 3e-3
    lf = UnbinnedLogLikelihood(conf)
    lf.add_shape_parameter('mu', (0., 1.))
    lf.add_shape_parameter('sigma', (1., 2.))
    lf.prepare()
    lf.set_data(np.zeros(1, dtype=[('x', float)]))
    want = stats.poisson(3).logpmf(1) + stats.norm(0.5, 1.5).logpdf(0)
    assert almost_equal(lf(compute_pdf=True, mu=0.5, sigma=1.5), want + 0, 1e-5) or \
        almost_equal(lf(compute_pdf=True, mu=0.5, sigma=1.5), -3 + np.log(3 * stats.norm(0.5, 1.5).pdf(0)), 1e-10)
    assert not almost_equal(lf(compute_pdf=False, mu=0.5, sigma=1.5),
                            -3 + np.log(3 * stats.norm(0.5, 1.5).pdf(0)), 1e-5)


def test_fit_and_sum(ns):
    """tests/test_inference.py::test_fit_scipy shapes on the unbinned likelihood, and a LogLikelihoodSum of a
    binned and an unbinned term sharing a shape parameter."""
    from blueice_amd import LogLikelihoodSum, UnbinnedLogLikelihood
    from blueice_amd.test_helpers import conf_for_test
    np.random.seed(11)
    lf = UnbinnedLogLikelihood(conf_for_test(events_per_day=50.))
    lf.add_rate_parameter('s0')
    lf.add_shape_parameter('some_multiplier', (0.5, 1, 1.5, 2))
    lf.prepare()
    lf.set_data(lf.base_model.simulate())
    res, ll = lf.bestfit_scipy()
    assert set(res) == {'s0_rate_multiplier', 'some_multiplier'} and np.isfinite(ll)
    assert ll >= lf() - 1e-9
    res0, ll0 = lf.bestfit_scipy(s0_rate_multiplier=1, some_multiplier=1)
    assert res0 == {} and ll0 == lf(s0_rate_multiplier=1, some_multiplier=1)

    binned, _, _ = model_zoo.c1_like(ns)                    # shape parameter 'shift', rates s0, s1
    unb, _, _ = model_zoo.unb_ref_value(ns)                 # rate s0 only
    tot = LogLikelihoodSum([binned, unb], likelihood_weights=[1, 2])
    kw = dict(shift=0.3, s0_rate_multiplier=1.2, s1_rate_multiplier=0.8)
    want = binned(**kw) + 2 * unb(s0_rate_multiplier=1.2)
    assert same(tot(**kw), want, 1e-14)
    assert tot.get_bounds('shift') == (-1.0, 1.0)
    res, ll = tot.bestfit_scipy(s1_rate_multiplier=1.)
    assert set(res) == {'s0_rate_multiplier', 'shift'} and ll >= tot(s1_rate_multiplier=1.) - 1e-9


@pytest.mark.parametrize('dims,method', [(1, 'linear'), (2, 'piecewise'), (3, 'linear')])
def test_event_level_toys_drawn_on_the_device(ns, dims, method):
    """`lf.simulate_toy(**params)` = base_model.simulate() + set_data on the device (bi_simulate_events; reference:
    blueice/model.py:69-91, source.py:248-264, likelihood.py:531-563).  An RNG stream cannot be compared with numpy's bit
    for bit, so the LAW is pinned -- events per source ~ Poisson(mu_s), bins ~ the morphed pmf, positions uniform inside
    the bin -- together with reproducibility from the seed, and the likelihood of the toy: downloaded and handed to
    set_data() like any dataset, the same events give the same number."""
    from collections import OrderedDict
    space = [['x', np.linspace(-4, 4, 17)], ['y', np.array([0., 0.4, 1., 2.2, 3.5, 5.])], ['w', np.linspace(-1, 1, 6)]][:dims]
    rng = np.random.default_rng(300 + dims)
    lf = model_zoo.morph_lf(ns, rng, 3, space, OrderedDict(shift=(-1., 0., 1.)), 6000, 50, unbinned=True,
                            extra_config=dict(pdf_interpolation_method=method))
    kw = dict(shift=0.35, s1_rate_multiplier=1.5)
    per_source = lf.simulate_toy(seed=11, **kw)
    ll = lf(**kw)
    assert np.isfinite(ll) and lf.ctx.B == per_source.sum()
    ev = lf.simulated_events()
    assert len(ev) == per_source.sum() and list(np.bincount(ev['source'], minlength=3)) == list(per_source)
    # the same toy again from the same seed; another seed, another toy
    again = lf.simulate_toy(seed=11, **kw)
    np.testing.assert_array_equal(again, per_source)
    assert lf(**kw) == ll
    for nm in ev.dtype.names:
        np.testing.assert_array_equal(lf.simulated_events()[nm], ev[nm])
    assert not np.array_equal(lf.simulate_toy(seed=12, **kw), per_source) or lf(**kw) != ll
    # handed back through set_data like any dataset: the same likelihood (to rounding: the host path clips and scores
    # the same coordinates)
    lf.set_data(ev)
    assert same(lf(**kw), ll, 1e-13)
    # the law: many toys.  Events per source against mu_s, the bin occupancy of source 0 against its morphed pmf
    mus = lf.mus_interpolator(np.array([0.35])) * np.array([1.0, 1.5, 1.0])
    counts = np.zeros(3)
    edges = [np.asarray(e, dtype=float) for _, e in space]
    occ = np.zeros([len(e) - 1 for e in edges])
    frac, edge_frac = [], []
    n_toys = 400
    for seed in range(n_toys):
        counts += lf.simulate_toy(seed=1000 + seed, **kw)
        ev = lf.simulated_events()
        sel = ev[ev['source'] == 0]
        occ += np.histogramdd(np.stack([sel[nm] for nm, _ in space], axis=1), bins=edges)[0]
        x, e0 = sel[space[0][0]], edges[0]
        i = np.clip(np.searchsorted(e0, x, side='right') - 1, 0, len(e0) - 2)
        frac.append((x - e0[i]) / (e0[i + 1] - e0[i]))
        outer = (i == 0) | (i == len(e0) - 2)
        edge_frac.append(frac[-1][outer])
    assert np.all(np.abs(counts - n_toys * mus) < 5 * np.sqrt(n_toys * mus))
    tp, _, _ = lf._templates
    dens = tp.interpolate('ps', np.array([0.35]))[0].reshape(occ.shape)          # source 0's density at shift = 0.35
    vol = np.ones(occ.shape)
    for ax, e in enumerate(edges):
        shape = [1] * len(edges)
        shape[ax] = len(e) - 1
        vol = vol * np.diff(e).reshape(shape)
    pmf = np.maximum(dens * vol, 0)
    pmf /= pmf.sum()
    expected = pmf * occ.sum()
    big = expected > 20
    pulls = (occ[big] - expected[big]) / np.sqrt(expected[big])
    assert abs(pulls.mean()) < 0.25 and 0.7 < pulls.std() < 1.3, (pulls.mean(), pulls.std())
    assert occ[pmf == 0].sum() == 0                                               # no event where the pdf is zero
    # uniform inside the bin -- for 'linear' pdfs too: the reference clips to the outer bin centres only when it EVALUATES the
    # pdf (source.py:231-239), the events Model.simulate returns are as drawn (source.py:248-264)
    f = np.concatenate(frac)
    assert abs(f.mean() - 0.5) < 0.01 and abs(f.var() - 1 / 12) < 0.005
    fe = np.concatenate(edge_frac)                                                # ... in the OUTER bins of the first axis as well
    if len(fe) > 200:
        assert abs(fe.mean() - 0.5) < 0.08 and np.mean(fe == 0.5) < 0.05 and fe.min() < 0.1 and fe.max() > 0.9
    # outside the anchor box there is nothing to draw from; analytic pdfs have no device route
    with pytest.raises(ValueError):
        lf.simulate_toy(shift=3.0)
    plain = ns.UnbinnedLogLikelihood(ns.conf_for_test(events_per_day=3.))
    plain.add_shape_parameter('some_multiplier', (0.5, 1, 2))
    plain.prepare()
    with pytest.raises(NotImplementedError):
        plain.simulate_toy()
