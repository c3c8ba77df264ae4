"""The likelihoods behind the golden fixtures, written once against a *namespace* of classes so the
very same construction code runs
  * on the real reference (tests/golden/make_golden.py, development container only) and
  * on blueice_amd (tests/test_dropin_*.py), whose results are then compared with the goldens.

`ns` must provide: BinnedLogLikelihood, DensityEstimatingSource, FixedSampleSource, GaussianMCSource,
conf_for_test, make_data.
"""
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np

BB_LC = {'model_statistical_uncertainty_handling': 'bb_single', 'bb_single_source': 0}


def namespace_of(package):
    """Namespace for `package` in {'blueice' (the reference), 'blueice_amd'}."""
    import importlib
    lk = importlib.import_module(package + '.likelihood')
    src = importlib.import_module(package + '.source')
    th = importlib.import_module(package + '.test_helpers')
    return SimpleNamespace(BinnedLogLikelihood=lk.BinnedLogLikelihood,
                           UnbinnedLogLikelihood=lk.UnbinnedLogLikelihood, GaussianSource=th.GaussianSource,
                           DensityEstimatingSource=src.DensityEstimatingSource,
                           FixedSampleSource=th.FixedSampleSource, GaussianMCSource=th.GaussianMCSource,
                           conf_for_test=th.conf_for_test, make_data=th.make_data,
                           LogLikelihoodSum=lk.LogLikelihoodSum, LogLikelihoodReParam=lk.LogLikelihoodReParam,
                           LogAncillaryLikelihood=lk.LogAncillaryLikelihood)


_morphed_cache = {}


def morphed_source_class(ns):
    """FixedSampleSource whose sample and rate respond to numeric shape settings: events are
    shifted / stretched / tilted with source-specific strengths."""
    key = ns.DensityEstimatingSource
    if key in _morphed_cache:
        return _morphed_cache[key]

    class MorphedSampleSource(ns.DensityEstimatingSource):
        def __init__(self, config, *args, **kwargs):
            super().__init__(config, *args, **kwargs)
            k = self.config.get('strength', 1.0)
            self.events_per_day *= (1 + 0.05 * k * self.config.get('shift', 0.)
                                    - 0.03 * k * self.config.get('stretch', 0.)
                                    + 0.02 * k * self.config.get('tilt', 0.))

        def get_events_for_density_estimate(self):
            d = self.config['data'].copy()
            k = self.config.get('strength', 1.0)
            names = [n for n, _ in self.config['analysis_space']]
            d[names[0]] = d[names[0]] + 0.31 * k * self.config.get('shift', 0.)
            if len(names) > 1:
                d[names[1]] = d[names[1]] * (1 + 0.11 * k * self.config.get('stretch', 0.))
            if len(names) > 2:
                d[names[2]] = d[names[2]] + 0.07 * k * self.config.get('tilt', 0.) * d[names[0]]
            return d, len(d)

    _morphed_cache[key] = MorphedSampleSource
    return MorphedSampleSource


def sample(rng, n, space):
    """n events inside the analysis space, clustered so that bins are unevenly filled."""
    d = np.zeros(n, dtype=[('source', int)] + [(nm, float) for nm, _ in space])
    for nm, edges in space:
        lo, hi = edges[0], edges[-1]
        u = rng.beta(2.0, 3.5, size=n)
        d[nm] = lo + (hi - lo) * (0.02 + 0.96 * u)
    return d


def morph_lf(ns, rng, S, space, shape_anchors, n_mc, n_data, lc=None, livetime=None, bb_floor=False, extra_config=None,
             unbinned=False):
    strengths = [1.0, -0.6, 0.45, 1.7, -1.2, 0.8]
    conf = dict(sources=[], default_source_class=morphed_source_class(ns),
                analysis_space=space, force_recalculation=True, never_save_to_cache=True,
                shift=0., stretch=0., tilt=0.)
    if livetime is not None:
        conf['livetime_days'] = livetime
    conf.update(extra_config or {})
    for s in range(S):
        d = sample(rng, n_mc, space)
        if bb_floor and s == 0:
            # >= 1 MC event in every bin for every anchor: add a lattice of bin centres
            centres = np.meshgrid(*[0.5 * (np.asarray(e)[1:] + np.asarray(e)[:-1]) for _, e in space], indexing='ij')
            lat = np.zeros(centres[0].size * 3, dtype=d.dtype)
            for (nm, _), c in zip(space, centres):
                lat[nm] = np.tile(c.ravel(), 3)
            d = np.concatenate([d, lat])
        conf['sources'].append(dict(name='s%d' % s, events_per_day=40. * (s + 1), data=d,
                                    strength=0.0 if (bb_floor and s == 0) else strengths[s]))
    lf = (ns.UnbinnedLogLikelihood if unbinned else ns.BinnedLogLikelihood)(conf, likelihood_config=dict(lc) if lc else None)
    for s in range(S):
        lf.add_rate_parameter('s%d' % s)
    for nm, anchors in shape_anchors.items():
        lf.add_shape_parameter(nm, anchors)
    lf.prepare()
    lf.set_data(sample(rng, n_data, space))
    return lf


# ---------------------------------------------------------------------------------------------
# cases: name -> function(ns) -> (lf, calls, full_output call indices)
# ---------------------------------------------------------------------------------------------
def ref_single_bin(ns):
    np.random.seed(1)
    lf = ns.BinnedLogLikelihood(ns.conf_for_test(mc=True, analysis_space=[['x', [-40, 40]]]))
    lf.add_rate_parameter('s0')
    lf.prepare()
    lf.set_data(np.zeros(1, dtype=[('x', float), ('source', int)]))
    return lf, [{}, dict(s0_rate_multiplier=5.4), dict(s0_rate_multiplier=0.)], (1,)


def ref_zero_bin(ns):
    np.random.seed(2)
    lf = ns.BinnedLogLikelihood(ns.conf_for_test(mc=True, analysis_space=[['x', [-40, 40]]]))
    lf.add_rate_parameter('s0')
    lf.prepare()
    lf.set_data(np.zeros(0, dtype=[('x', float), ('source', int)]))
    return lf, [dict(s0_rate_multiplier=0.), {}, dict(s0_rate_multiplier=2.)], ()


def ref_multi_bin_single_dim(ns):
    data, _ = ns.make_data([dict(n_events=24, x=0.5), dict(n_events=56, x=1.5)])
    conf = ns.conf_for_test(events_per_day=42, analysis_space=[['x', [0, 1, 5]]],
                            default_source_class=ns.FixedSampleSource, data=data)
    lf = ns.BinnedLogLikelihood(conf)
    lf.add_rate_parameter('s0')
    data, _ = ns.make_data([dict(n_events=18, x=0.5), dict(n_events=70, x=1.5)])
    lf.set_data(data)          # no prepare(): the auto-prepare path
    return lf, [{}, dict(s0_rate_multiplier=1.7)], ()


def ref_multi_bin(ns):
    data, _ = ns.make_data([dict(n_events=24, x=0.5, y=0.5), dict(n_events=56, x=1.5, y=0.5),
                            dict(n_events=6, x=0.5, y=2), dict(n_events=14, x=1.5, y=2)])
    conf = ns.conf_for_test(events_per_day=42, default_source_class=ns.FixedSampleSource, data=data,
                            analysis_space=[['x', [0, 1, 5]], ['y', [0, 1, 4]]])
    lf = ns.BinnedLogLikelihood(conf)
    lf.add_rate_parameter('s0')
    lf.add_shape_parameter('strlen_multiplier', {1: 'x', 2: 'hi', 3: 'wha'}, base_value=1)
    lf.prepare()
    data, _ = ns.make_data([dict(n_events=18, x=0.5, y=0.5), dict(n_events=70, x=1.5, y=0.5),
                            dict(n_events=4, x=0.5, y=2), dict(n_events=10, x=1.5, y=2)])
    lf.set_data(data)
    calls = [dict(strlen_multiplier=1), dict(strlen_multiplier=2), dict(strlen_multiplier=2.3),
             dict(strlen_multiplier=3), dict(strlen_multiplier=1.00001, s0_rate_multiplier=0.3),
             dict(strlen_multiplier=3.5), dict(strlen_multiplier=0.99)]
    return lf, calls, (2,)


def ref_bb_single_bin(ns):
    data, _ = ns.make_data([dict(n_events=32, x=0.5)])
    conf = ns.conf_for_test(default_source_class=ns.FixedSampleSource, events_per_day=32 / 5,
                            analysis_space=[['x', [0, 1]]], data=data)
    lf = ns.BinnedLogLikelihood(conf, likelihood_config=dict(BB_LC))
    lf.prepare()
    lf.set_data(np.zeros(2, dtype=[('x', float), ('source', int)]))
    return lf, [{}], (0,)


def ref_bb_multi_bin(ns):
    data, _ = ns.make_data([dict(n_events=16, x=0.5), dict(n_events=30, x=1.5),
                            dict(n_events=32, x=2.5), dict(n_events=27, x=3.5)])
    conf = ns.conf_for_test(default_source_class=ns.FixedSampleSource, events_per_day=105 / 5,
                            analysis_space=[['x', [0, 1, 2, 3, 4]]], data=data)
    lf = ns.BinnedLogLikelihood(conf, likelihood_config=dict(BB_LC))
    lf.prepare()
    data, _ = ns.make_data([dict(n_events=3, x=0.5), dict(n_events=5, x=1.5),
                            dict(n_events=2, x=2.5), dict(n_events=7, x=3.5)])
    lf.set_data(data)
    return lf, [{}], (0,)


def _bb_second_source_lf(ns, extra=False):
    cal, _ = ns.make_data([dict(n_events=16, x=0.5), dict(n_events=30, x=1.5),
                           dict(n_events=32, x=2.5), dict(n_events=27, x=3.5)])
    oth, _ = ns.make_data([dict(n_events=5, x=0.5), dict(n_events=7, x=1.5),
                           dict(n_events=1, x=2.5), dict(n_events=3, x=3.5)])
    conf = ns.conf_for_test(default_source_class=ns.FixedSampleSource,
                            analysis_space=[['x', [0, 1, 2, 3, 4]]], dummy=1)
    conf['sources'] = [{'name': 's0', 'events_per_day': 105 / 5., 'data': cal},
                       {'name': 's1', 'events_per_day': 16., 'data': oth}]
    lf = ns.BinnedLogLikelihood(conf, likelihood_config=dict(BB_LC))
    lf.add_shape_parameter('dummy', (0, 1))
    if extra:
        lf.add_rate_parameter('s1')
        lf.add_rate_parameter('s0')
        lf.add_shape_parameter('strlen_multiplier', {1: 'x', 2: 'hi', 3: 'wha'}, base_value=1)
    lf.prepare()
    data, _ = ns.make_data([dict(n_events=3, x=0.5), dict(n_events=5, x=1.5),
                            dict(n_events=2, x=2.5), dict(n_events=7, x=3.5)])
    lf.set_data(data)
    return lf


def ref_bb_second_source(ns):
    return _bb_second_source_lf(ns), [{}, dict(dummy=0.25), dict(dummy=0)], (0,)


def bb_two_shape(ns):
    calls = [{}, dict(strlen_multiplier=1.5, dummy=.25),
             dict(strlen_multiplier=2.3, dummy=.5, s1_rate_multiplier=.8),
             dict(strlen_multiplier=3, dummy=1, s0_rate_multiplier=1.3), dict(strlen_multiplier=2, dummy=0.999),
             dict(s1_rate_multiplier=0.),
             dict(strlen_multiplier=2.75, dummy=0.1, s0_rate_multiplier=0.4, s1_rate_multiplier=2.2)]
    return _bb_second_source_lf(ns, extra=True), calls, (2, 5)


def c1_like(ns):
    rng = np.random.default_rng(11)
    space = [['x', np.linspace(-4, 4, 41)]]
    lf = morph_lf(ns, rng, 2, space, OrderedDict(shift=(-1., 0., 1.)), 4000, 300)
    calls = [{}, dict(shift=-1.), dict(shift=1.), dict(shift=0.37), dict(shift=-0.82, s0_rate_multiplier=1.4),
             dict(shift=0.999999, s1_rate_multiplier=0.), dict(shift=1.2), dict(shift=-1.0000001),
             dict(s0_rate_multiplier=-0.1), dict(shift=0.5, s0_rate_multiplier=0., s1_rate_multiplier=0.)]
    return lf, calls, (3,)


def d2_nonuniform(ns):
    rng = np.random.default_rng(12)
    space = [['x', np.array([-3., -1.5, -0.5, 0., 0.4, 1.1, 3.])], ['y', np.linspace(0, 5, 6)]]
    lf = morph_lf(ns, rng, 3, space, OrderedDict(shift=(-1., -0.25, 0.5, 2.), stretch=(0., 1., 4.)),
                  3000, 500, livetime=2.)
    calls = [{}]
    for z0 in (-1., -0.6, -0.25, 0.1, 0.5, 1.3, 2.):
        for z1 in (0., 0.5, 1., 2.5, 4.):
            calls.append(dict(shift=z0, stretch=z1))
    calls += [dict(shift=0.3, stretch=3.3, s0_rate_multiplier=0.5, s1_rate_multiplier=2., s2_rate_multiplier=1.1),
              dict(shift=0.3, stretch=3.3, livetime_days=5.), dict(shift=2.01, stretch=1.),
              dict(shift=0., stretch=-0.01), dict(shift=1., stretch=2., s2_rate_multiplier=-1.)]
    return lf, calls, (9, 36)


def d3_small(ns):
    rng = np.random.default_rng(13)
    space = [['x', np.linspace(-3, 3, 7)], ['y', np.linspace(0, 5, 6)], ['w', np.linspace(-2, 2, 5)]]
    lf = morph_lf(ns, rng, 4, space,
                  OrderedDict(shift=(-1., 0., 1.), stretch=(-1., 0., 1.), tilt=(-1., 0., 1.)), 5000, 800)
    calls = [{}]
    pts = rng.uniform(-1, 1, size=(16, 3))
    special = [(-1, -1, -1), (1, 1, 1), (0, 0, 0), (1, -1, 0), (0.5, 1, -1), (-1, 0.25, 1), (1, 0, 0.75),
               (0, 0, 1e-9), (-1e-12, 0, 0)]
    for p in list(pts) + [np.array(s, dtype=float) for s in special]:
        calls.append(dict(shift=float(p[0]), stretch=float(p[1]), tilt=float(p[2])))
    calls += [dict(shift=.2, stretch=-.7, tilt=.4, s0_rate_multiplier=1.2, s1_rate_multiplier=0.,
                   s2_rate_multiplier=0.7, s3_rate_multiplier=3.),
              dict(shift=1.0000001, stretch=0, tilt=0), dict(shift=float('nan'), stretch=0, tilt=0)]
    return lf, calls, (5, 26)


def d0_multi_source(ns):
    rng = np.random.default_rng(14)
    space = [['x', np.linspace(-3, 3, 13)], ['y', np.linspace(0, 5, 4)]]
    lf = morph_lf(ns, rng, 3, space, OrderedDict(), 2000, 150)
    return lf, [{}, dict(s0_rate_multiplier=2.), dict(s1_rate_multiplier=0., s2_rate_multiplier=.5)], (1,)


def _edge_lf(ns):
    data, _ = ns.make_data([dict(n_events=10, x=0.5), dict(n_events=30, x=2.5)])
    conf = ns.conf_for_test(events_per_day=20, analysis_space=[['x', [0, 1, 2, 3]]],
                            default_source_class=ns.FixedSampleSource, data=data)
    lf = ns.BinnedLogLikelihood(conf)
    lf.add_rate_parameter('s0')
    lf.add_shape_parameter('strlen_multiplier', {1: 'x', 2: 'hi'}, base_value=1)
    lf.prepare()
    return lf


_EDGE_CALLS = [{}, dict(strlen_multiplier=1.5), dict(s0_rate_multiplier=0.)]


def edge_mu_zero_hit(ns):
    lf = _edge_lf(ns)
    lf.set_data(ns.make_data([dict(n_events=3, x=0.5), dict(n_events=1, x=1.5)])[0])
    return lf, list(_EDGE_CALLS), ()


def edge_mu_zero_ok(ns):
    lf = _edge_lf(ns)
    lf.set_data(ns.make_data([dict(n_events=3, x=0.5), dict(n_events=11, x=2.5)])[0])
    return lf, list(_EDGE_CALLS), ()


def edge_empty_data(ns):
    lf = _edge_lf(ns)
    lf.set_data(ns.make_data([])[0])
    return lf, [{}, dict(strlen_multiplier=2), dict(s0_rate_multiplier=0.)], ()


def bb_d2(ns):
    rng = np.random.default_rng(15)
    space = [['x', np.linspace(-3, 3, 7)], ['y', np.linspace(0, 5, 4)]]
    lf = morph_lf(ns, rng, 3, space, OrderedDict(shift=(-1., 0., 1.), stretch=(0., 1.)), 2500, 400,
                  lc=BB_LC, bb_floor=True)
    calls = [{}]
    for p in rng.uniform(0, 1, size=(10, 2)):
        calls.append(dict(shift=float(2 * p[0] - 1), stretch=float(p[1])))
    calls += [dict(shift=-1., stretch=1.), dict(shift=.4, stretch=.6, s0_rate_multiplier=1.5, s1_rate_multiplier=.2),
              dict(shift=.4, stretch=.6, s1_rate_multiplier=0., s2_rate_multiplier=0.)]
    return lf, calls, (4, 12)


def neg_allowed(ns):
    """A source that may have a negative rate (config 'allow_negative', likelihood.py:82-83,397-415): finite
    while every bin's total stays positive, nan once a bin goes negative, -inf when the summed rate does."""
    rng = np.random.default_rng(41)
    space = [['x', np.linspace(-3, 3, 9)]]
    conf = dict(sources=[dict(name='bkg', events_per_day=200., data=sample(rng, 3000, space), strength=0.3),
                         dict(name='sig', events_per_day=20., data=sample(rng, 1500, space), strength=-1.0,
                              allow_negative=True)],
                default_source_class=morphed_source_class(ns), analysis_space=space,
                force_recalculation=True, never_save_to_cache=True, shift=0., stretch=0., tilt=0.)
    lf = ns.BinnedLogLikelihood(conf)
    lf.add_rate_parameter('bkg')
    lf.add_rate_parameter('sig')
    lf.add_shape_parameter('shift', (-1., 0., 1.))
    lf.prepare()
    lf.set_data(sample(rng, 230, space))
    calls = [{}, dict(sig_rate_multiplier=-0.5), dict(sig_rate_multiplier=-2., shift=0.4),
             dict(sig_rate_multiplier=-9.), dict(sig_rate_multiplier=-30.), dict(bkg_rate_multiplier=-0.1),
             dict(sig_rate_multiplier=-1., bkg_rate_multiplier=0.), dict(shift=-0.7, sig_rate_multiplier=3.)]
    return lf, calls, (1,)


def efficiency_param(ns):
    """Sources scaled by an efficiency that is itself a shape parameter (likelihood.py:84-87,385-393)."""
    rng = np.random.default_rng(42)
    space = [['x', np.linspace(-3, 3, 7)], ['y', np.linspace(0, 5, 4)]]
    conf = dict(sources=[dict(name='a', events_per_day=50., data=sample(rng, 2000, space), strength=1.0),
                         dict(name='b', events_per_day=30., data=sample(rng, 2000, space), strength=-0.5,
                              apply_efficiency=True, efficiency_name='eff'),
                         dict(name='c', events_per_day=10., data=sample(rng, 2000, space), strength=0.2,
                              apply_efficiency=True, efficiency_name='eff2')],
                default_source_class=morphed_source_class(ns), analysis_space=space,
                force_recalculation=True, never_save_to_cache=True, shift=0., stretch=0., tilt=0., eff=1.0)
    lf = ns.BinnedLogLikelihood(conf)
    for n in 'abc':
        lf.add_rate_parameter(n)
    lf.add_shape_parameter('shift', (-1., 0., 1.))
    lf.add_shape_parameter('eff', (0.5, 1.0, 1.5))
    lf.prepare()
    lf.set_data(sample(rng, 120, space))
    calls = [{}, dict(eff=0.5), dict(eff=1.3, shift=0.2), dict(eff=0.8, b_rate_multiplier=2., c_rate_multiplier=0.5),
             dict(eff=1.5, shift=-1., a_rate_multiplier=0.)]
    return lf, calls, (2,)


def fit_c1_like(ns):
    rng = np.random.default_rng(21)
    space = [['x', np.linspace(-4, 4, 41)]]
    return morph_lf(ns, rng, 2, space, OrderedDict(shift=(-1., 0., 1.)), 4000, 300)


def profile_d2(ns):
    """Three sources, two shape parameters (non-uniform anchors), 2-d space: profiled scans with FOUR floating nuisances
    (two rates, two shapes) that cross grid-cell boundaries of the morph."""
    rng = np.random.default_rng(31)
    space = [['x', np.linspace(-3, 3, 13)], ['y', np.linspace(0, 5, 9)]]
    return morph_lf(ns, rng, 3, space, OrderedDict(shift=(-1., -0.25, 0.5, 2.), stretch=(0., 1., 4.)), 6000, 700)


# the profiled scans whose per-point maxima tests/golden/make_golden_profile.py records from the reference's own
# bestfit_scipy (blueice/inference.py:131-178 under the loops of :392-443): name -> (builder, scan space, fixed kwargs)
PROFILE_SCANS = OrderedDict([
    ('c1_shift_1001', (fit_c1_like, [('shift', np.linspace(-1., 1., 1001))], {})),
    ('c1_rate_x_shift_24x24', (fit_c1_like, [('s0_rate_multiplier', np.linspace(0.2, 3., 24)), ('shift', np.linspace(-0.95, 0.95, 24))], {})),
    ('d2_rate_160', (profile_d2, [('s0_rate_multiplier', np.linspace(0.05, 4., 160))], {'s2_rate_multiplier': 1.})),
    # the same with the second shape parameter held inside its range: in the scan above its best value sits ON the
    # lower anchor, where the reference's minimisers meet an infinite wall and mostly stop short of the maximum
    ('d2_rate_120_interior', (profile_d2, [('s0_rate_multiplier', np.linspace(0.05, 4., 120))],
                              {'s2_rate_multiplier': 1., 'stretch': 1.5})),
])


# ---------------------------------------------------------------------------------------------
# unbinned cases: analytic Gaussian sources scored at fixed events
# ---------------------------------------------------------------------------------------------
def _events(xs):
    d = np.zeros(len(xs), dtype=[('x', float), ('source', int)])
    d['x'] = xs
    return d


def unb_ref_value(ns):
    """tests/test_likelihood.py::test_likelihood_value of the reference."""
    lf = ns.UnbinnedLogLikelihood(ns.conf_for_test(events_per_day=1))
    lf.add_rate_parameter('s0')
    lf.set_data(_events([0.]))
    return lf, [{}, dict(s0_rate_multiplier=2), dict(s0_rate_multiplier=0.)], (1,)


def unb_shape_2src(ns):
    rng = np.random.default_rng(31)
    conf = ns.conf_for_test(n_sources=2, events_per_day=15., livetime_days=2.)
    conf['sources'] = [dict(name='s0', mu=-0.5), dict(name='s1', mu=1.0, sigma=0.7, events_per_day=9.)]
    lf = ns.UnbinnedLogLikelihood(conf)
    lf.add_rate_parameter('s0')
    lf.add_rate_parameter('s1')
    lf.add_shape_parameter('sigma', (0.6, 1., 1.5, 2.5))
    lf.add_shape_parameter('some_multiplier', (0.5, 1, 2))
    lf.prepare()
    xs = np.concatenate([rng.normal(-0.5, 1.1, 30), rng.normal(1.0, 0.7, 15), [60., -75.]])   # two far outliers
    lf.set_data(_events(xs))
    calls = [{}, dict(sigma=0.6), dict(sigma=2.5, some_multiplier=2), dict(sigma=1.2, some_multiplier=0.7),
             dict(sigma=0.9, some_multiplier=1.6, s0_rate_multiplier=1.4, s1_rate_multiplier=0.3),
             dict(sigma=1.0, s0_rate_multiplier=0., s1_rate_multiplier=0.), dict(sigma=2.6),
             dict(sigma=1.3, livetime_days=5.), dict(s1_rate_multiplier=-1.)]
    return lf, calls, (3,)


def unb_d0_three_sources(ns):
    rng = np.random.default_rng(32)
    conf = ns.conf_for_test(n_sources=3, events_per_day=5.)
    conf['sources'] = [dict(name='a', mu=-2.), dict(name='b', mu=0., sigma=2.), dict(name='c', mu=3., events_per_day=1.)]
    lf = ns.UnbinnedLogLikelihood(conf)
    lf.add_rate_parameter('a')
    lf.add_rate_parameter('c')
    lf.set_data(_events(rng.normal(0, 2.5, 25)))
    return lf, [{}, dict(a_rate_multiplier=3.), dict(a_rate_multiplier=0., c_rate_multiplier=2.)], (0,)


def unb_no_events(ns):
    lf = ns.UnbinnedLogLikelihood(ns.conf_for_test(events_per_day=3.))
    lf.add_rate_parameter('s0')
    lf.add_shape_parameter('some_multiplier', (0.5, 1, 2))
    lf.prepare()
    lf.set_data(_events([]))
    return lf, [{}, dict(some_multiplier=1.5, s0_rate_multiplier=2.)], ()


_holey_cache = {}


def holey_gaussian_class(ns):
    """GaussianSource whose pdf is nan beyond `nan_beyond` sigmas (a density estimate that gives up in its tails):
    extended_loglikelihood drops such terms with np.nansum (blueice/likelihood.py:686)."""
    key = ns.GaussianSource
    if key not in _holey_cache:
        class HoleyGaussianSource(ns.GaussianSource):
            def pdf(self, *args):
                p = np.asarray(super().pdf(*args), dtype=float).copy()
                x = np.asarray(args[0], dtype=float)
                cut = self.config.get('nan_beyond')
                if cut is not None:
                    p[np.abs(x - self.config['mu']) > cut * self.config['sigma']] = np.nan
                return p
        _holey_cache[key] = HoleyGaussianSource
    return _holey_cache[key]


def unb_nan_pdf(ns):
    """Sources whose pdf is nan at some events, for some anchors only: np.nansum semantics of the unbinned
    likelihood (an event all of whose sources are nan gets the outlier likelihood)."""
    rng = np.random.default_rng(33)
    conf = ns.conf_for_test(n_sources=3, events_per_day=12., default_source_class=holey_gaussian_class(ns))
    conf['sources'] = [dict(name='a', mu=-1., nan_beyond=2.0), dict(name='b', mu=1.5, sigma=0.8, nan_beyond=2.5),
                       dict(name='c', mu=0., sigma=3., events_per_day=4.)]
    lf = ns.UnbinnedLogLikelihood(conf)
    for n in 'abc':
        lf.add_rate_parameter(n)
    lf.add_shape_parameter('sigma', (0.7, 1., 1.6))
    lf.prepare()
    xs = np.concatenate([rng.normal(-1., 1.2, 25), rng.normal(1.5, 0.9, 15), [-4.4, 5.9, 0.2, 9.5, -9.]])
    lf.set_data(_events(xs))
    calls = [{}, dict(sigma=0.7), dict(sigma=1.6), dict(sigma=1.3), dict(sigma=0.85, a_rate_multiplier=2., c_rate_multiplier=0.),
             dict(sigma=1.0, b_rate_multiplier=0.), dict(sigma=1.45, c_rate_multiplier=0., b_rate_multiplier=1.7)]
    return lf, calls, (3,)


def unb_mc_hist(ns):
    """Sources whose pdf is a HISTOGRAM of their own Monte Carlo, read with the default linear interpolation between
    bin centres (blueice/source.py:218-243): the set_data the device does itself (bi_score_events).  Events on bin
    centres, on edges, in the outer half of the boundary bins and on the range limits."""
    np.random.seed(7)
    rng = np.random.default_rng(71)
    conf = ns.conf_for_test(n_sources=2, mc=True, events_per_day=30., n_events_for_pdf=int(1e5),
                            analysis_space=[['x', np.linspace(-6, 6, 49)]],
                            force_recalculation=True, never_save_to_cache=True)   # no pdf cache: every source draws its own sample
    conf['sources'] = [dict(name='s0'), dict(name='s1', mu=1.0, sigma=1.8, events_per_day=12.)]
    lf = ns.UnbinnedLogLikelihood(conf)
    lf.add_rate_parameter('s0')
    lf.add_rate_parameter('s1')
    lf.add_shape_parameter('mu', (-1., 0., 1.5))
    lf.prepare()
    xs = np.concatenate([rng.normal(0.2, 1.2, 40), [-6., 6., -5.9, 5.95, 0.125, 0.25, -0.375, 3.0]])
    lf.set_data(_events(xs))
    calls = [{}, dict(mu=-1.), dict(mu=1.5), dict(mu=0.6), dict(mu=-0.45, s0_rate_multiplier=1.3, s1_rate_multiplier=0.5),
             dict(mu=1.1, s1_rate_multiplier=0.)]
    return lf, calls, (3,)


UNBINNED_CASES = OrderedDict((f.__name__, f) for f in (unb_ref_value, unb_shape_2src, unb_d0_three_sources, unb_no_events,
                                                       unb_nan_pdf, unb_mc_hist))


# ---------------------------------------------------------------------------------------------
# API cases: calls whose result is not a function of the anchor tensors alone (compute_pdf builds a model at the
# point; priors; raised errors).  Goldens hold the reference's return values / exception names only
# (tests/golden/api_<name>.npz) and are compared through the drop-in API on the GPU.
# ---------------------------------------------------------------------------------------------
def api_compute_pdf(ns):
    """compute_pdf=True in and OUTSIDE the anchor box, with a shape prior and a rate prior: the reference builds the
    model at the point, applies no bounds test and adds only the rate priors (blueice/likelihood.py:331-335,366-371)."""
    from scipy import stats
    rng = np.random.default_rng(51)
    space = [['x', np.linspace(-4, 4, 17)], ['y', np.linspace(0, 5, 6)]]
    lf = morph_lf(ns, rng, 2, space, OrderedDict(shift=(-1., 0., 1.), stretch=(0., 1.)), 3000, 250)
    lf.rate_parameters['s1'] = stats.norm(1, 0.3).logpdf
    lf.shape_parameters['shift'] = (lf.shape_parameters['shift'][0], stats.norm(0, 0.5).logpdf, None)
    calls = [dict(shift=0.4, stretch=0.3), dict(compute_pdf=True, shift=0.4, stretch=0.3),
             dict(compute_pdf=True, shift=1., stretch=1.), dict(shift=1., stretch=1.),
             dict(compute_pdf=True, shift=1.4, stretch=0.5),                      # outside the box: finite
             dict(shift=1.4, stretch=0.5),                                        # ... -inf when interpolating
             dict(compute_pdf=True, shift=-1.7, stretch=1.3, s1_rate_multiplier=0.8, s0_rate_multiplier=1.2),
             dict(compute_pdf=True, shift=0.2, stretch=0.9, s1_rate_multiplier=-0.5),   # unphysical
             dict(compute_pdf=True)]
    return lf, calls


def api_livetime_zero(ns):
    """Base live time 0 (blueice/likelihood.py:374-380): scaling 0 -> 0 is allowed (all rates are 0), 0 -> non-0
    raises ValueError."""
    rng = np.random.default_rng(52)
    space = [['x', np.linspace(-3, 3, 7)]]
    lf = morph_lf(ns, rng, 2, space, OrderedDict(shift=(-1., 0., 1.)), 1500, 0, livetime=0.)
    calls = [dict(livetime_days=0.), dict(livetime_days=0., shift=0.3, s0_rate_multiplier=2.), dict(shift=0.5),
             dict(livetime_days=2.), dict(livetime_days=0., shift=1.5)]
    return lf, calls


def api_livetime_zero_with_data(ns):
    rng = np.random.default_rng(53)
    space = [['x', np.linspace(-3, 3, 7)]]
    lf = morph_lf(ns, rng, 2, space, OrderedDict(shift=(-1., 0., 1.)), 1500, 40, livetime=0.)
    return lf, [dict(livetime_days=0.), dict(shift=-0.2), dict(livetime_days=1e-3)]


def api_source_wise(ns):
    """Source-wise interpolation of an unbinned likelihood (blueice/likelihood.py:152-171,210-240,534-563): source a
    responds to mu only, b to sigma only, c to neither (its settings are fixed in its own spec); calls on and off the
    anchors, with rate multipliers, and outside the box."""
    rng = np.random.default_rng(61)
    conf = ns.conf_for_test(n_sources=3, events_per_day=20., source_wise_interpolation=True)
    conf['sources'] = [dict(name='a', extra_dont_hash_settings=['sigma']),
                       dict(name='b', mu=1.5, extra_dont_hash_settings=['mu']),
                       dict(name='c', mu=-2., sigma=2.5, events_per_day=5., extra_dont_hash_settings=['mu', 'sigma'])]
    lf = ns.UnbinnedLogLikelihood(conf)
    for n in 'abc':
        lf.add_rate_parameter(n)
    lf.add_shape_parameter('mu', (-1., 0., 1.))
    lf.add_shape_parameter('sigma', (0.8, 1., 1.5))
    lf.prepare()
    xs = np.concatenate([rng.normal(0.2, 1.1, 30), rng.normal(1.5, 1.2, 25), rng.normal(-2., 2.5, 8)])
    lf.set_data(_events(xs))
    calls = [{}, dict(mu=0.5), dict(sigma=1.2), dict(mu=-0.7, sigma=0.9), dict(mu=1., sigma=1.5),
             dict(mu=0.3, sigma=1.35, a_rate_multiplier=1.7, b_rate_multiplier=0.4, c_rate_multiplier=0.),
             dict(mu=-1., sigma=0.8, c_rate_multiplier=2.), dict(mu=1.2), dict(sigma=0.5)]
    return lf, calls


def api_source_wise_binned(ns):
    """The binned likelihood refuses source-wise interpolation in prepare() (blueice/likelihood.py:590-591)."""
    conf = ns.conf_for_test(n_sources=2, source_wise_interpolation=True)
    lf = ns.BinnedLogLikelihood(conf)
    lf.add_shape_parameter('mu', (-1., 0., 1.))

    class _Prepare:                       # dump_api / the parity test call lf(**kw): make prepare() the call
        def __call__(self, **kw):
            lf.prepare()
            return 0.
    return _Prepare(), [{}]


def _conv_config():
    return dict(np0=(np.linspace(1e-12, 10, 2), None, None), np1=(np.linspace(1e-12, 10, 2), None, None),
                op0_rate_multiplier=dict(params=['np0'], func=lambda a: a ** 2),
                op1_rate_multiplier=dict(params=['np1'], func=lambda b: b ** 2),
                op2_rate_multiplier=dict(params=['np0', 'np1'], func=lambda a, b: a * b))


def api_reparam_unbinned(ns):
    """LogLikelihoodReParam over an unbinned likelihood of three sources (blueice/likelihood.py:715-864; the set-up of
    the reference's tests/test_likelihood_reparam.py), with data that is not all at x = 0."""
    rng = np.random.default_rng(62)
    conf = ns.conf_for_test(events_per_day=1.)
    conf['sources'] = [dict(name='op0'), dict(name='op1', mu=0.8), dict(name='op2', sigma=1.7)]
    conf['np0'] = conf['np1'] = 1
    inner = ns.UnbinnedLogLikelihood(conf)
    for n in ('op0', 'op1', 'op2'):
        inner.add_rate_parameter(n)
    inner.prepare()
    lf = ns.LogLikelihoodReParam(inner, _conv_config())
    lf.set_data(_events(rng.normal(0.3, 1.3, 12)))
    calls = [{}, dict(np0=2.), dict(np1=3.), dict(np0=2., np1=2.), dict(np0=0.5, np1=1.5), dict(np0=1e-12, np1=10.)]
    return lf, calls


def api_reparam_binned(ns):
    """LogLikelihoodReParam over a BINNED likelihood with a shape parameter: the new parameters become rate multipliers,
    the shape parameter passes through to the morph."""
    from scipy import stats
    rng = np.random.default_rng(63)
    space = [['x', np.linspace(-4, 4, 17)], ['y', np.linspace(0, 5, 6)]]
    inner = morph_lf(ns, rng, 2, space, OrderedDict(shift=(-1., 0., 1.)), 3000, 250, extra_config=dict(amp=1., tilt2=2.))
    inner.rate_parameters['s0'] = stats.norm(1, 0.4).logpdf
    conv = dict(amp=((0.5, 2.), None, None), tilt2=((1., 4.), None, None),
                s0_rate_multiplier=dict(params=['amp'], func=lambda a: a ** 3),
                s1_rate_multiplier=dict(params=['amp', 'tilt2'], func=lambda a, t: a + 0.5 * t))
    lf = ns.LogLikelihoodReParam(inner, conv)
    calls = [{}, dict(shift=0.4), dict(amp=1.5), dict(amp=0.7, tilt2=3., shift=-0.6), dict(tilt2=1., shift=1.),
             dict(amp=2., tilt2=4., shift=1.3)]
    return lf, calls


def api_sum_ancillary(ns):
    """LogLikelihoodSum of a binned likelihood and an analytic constraint (LogAncillaryLikelihood,
    blueice/likelihood.py:867-1001), weighted."""
    from scipy import stats
    rng = np.random.default_rng(64)
    space = [['x', np.linspace(-4, 4, 17)]]
    binned = morph_lf(ns, rng, 2, space, OrderedDict(shift=(-1., 0., 1.)), 2000, 120)

    def constraint(values, width):
        return stats.norm.logpdf(values['shift'], 0.1, width) - 0.5 * (values['nuisance'] / 2.) ** 2

    anc = ns.LogAncillaryLikelihood(constraint, ['shift', 'nuisance'], config=dict(shift=0., nuisance=0.7),
                                    func_kwargs=dict(width=0.3))
    lf = ns.LogLikelihoodSum([binned, anc], likelihood_weights=[1, 0.5])
    calls = [{}, dict(shift=0.35), dict(shift=-0.8, nuisance=-1.2), dict(shift=0.5, s1_rate_multiplier=1.4, nuisance=0.),
             dict(shift=1.2, nuisance=3.)]
    return lf, calls


API_CASES = OrderedDict((f.__name__, f) for f in (api_compute_pdf, api_livetime_zero, api_livetime_zero_with_data,
                                                  api_source_wise, api_source_wise_binned, api_reparam_unbinned,
                                                  api_reparam_binned, api_sum_ancillary))

CASES = OrderedDict((f.__name__, f) for f in (
    ref_single_bin, ref_zero_bin, ref_multi_bin_single_dim, ref_multi_bin, ref_bb_single_bin,
    ref_bb_multi_bin, ref_bb_second_source, bb_two_shape, c1_like, d2_nonuniform, d3_small,
    d0_multi_source, edge_mu_zero_hit, edge_mu_zero_ok, edge_empty_data, bb_d2, neg_allowed, efficiency_param))
