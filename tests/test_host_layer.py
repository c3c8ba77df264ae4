"""Host logic of the drop-in (no GPU): the Model / Source / Histdd / morpher / likelihood plumbing must
hand the device exactly the tensors the reference builds and
implement the scalar bookkeeping of row a4.  The device is replaced by a recorder -- nothing is
computed in its place."""
from collections import OrderedDict

import numpy as np
import pytest

import model_zoo
from golden_util import load_case


class RecordingContext:
    """Stands where DeviceContext would: records uploads, cannot evaluate."""
    instances = []

    def __init__(self, device=None):
        self.device = device
        self.anchors = {}
        self.counts = None
        self.T = 0
        RecordingContext.instances.append(self)

    def begin_model(self, anchor_z, S, B, bb_source=-1):
        self.anchor_z, self.S, self.B, self.bb_source = [np.asarray(g, float) for g in anchor_z], S, B, bb_source
        self.anchors = {}

    def set_anchor(self, idx, ps, mus, n_model_row=None):
        self.anchors[idx] = (np.array(ps, float).reshape(self.S, self.B), np.array(mus, float),
                             None if n_model_row is None else np.array(n_model_row, float).reshape(self.B))

    def end_model(self):
        pass

    def set_allow_negative(self, flags):
        self.allow_negative = list(flags)

    def upload_counts(self, counts):
        self.counts = np.array(counts, float)
        self.T = self.counts.size // self.B

    def set_analysis_space(self, edges):
        self.edges = [np.asarray(e, float) for e in edges]

    def upload_events(self, *coords):
        # the recorder bins on the host only to let the test compare with the reference's counts; the
        # device kernel that does it for real is pinned against numpy.histogramdd in tests/test_gpu_golden.py
        sample = np.stack([np.asarray(c, float).ravel() for c in coords], axis=1)
        self.counts = np.histogramdd(sample, bins=self.edges)[0] if len(sample) else np.zeros([len(e) - 1 for e in self.edges])
        self.T = 1

    def download_counts(self, t=0):
        return np.array(self.counts, float).ravel()

    def set_unbinned(self, outlier):
        self.outlier = outlier
        self.T = 1

    def close(self):
        pass


@pytest.fixture()
def ns(monkeypatch):
    import blueice_amd.likelihood as lk
    monkeypatch.setattr(lk, 'DeviceContext', RecordingContext)
    RecordingContext.instances.clear()
    return model_zoo.namespace_of('blueice_amd')


@pytest.mark.parametrize('name', list(model_zoo.CASES))
def test_uploaded_tensors_equal_reference_tensors(ns, name):
    lf, calls, _ = model_zoo.CASES[name](ns)
    c = load_case(name)
    rec = lf.ctx
    assert isinstance(rec, RecordingContext)
    assert rec.S == c['S'] and rec.B == int(np.prod(c['bins'])) and rec.bb_source == c['bb_source']
    assert len(rec.anchor_z) == c['d']
    for g, g_ref in zip(rec.anchor_z, c['model']['anchor_z']):
        np.testing.assert_array_equal(g, g_ref)
    grid_shape = tuple(len(g) for g in c['model']['anchor_z'])
    ps_ref = c['model']['ps'].reshape((-1, c['S'], rec.B))
    mus_ref = c['model']['mus'].reshape((-1, c['S']))
    assert len(rec.anchors) == int(np.prod(grid_shape)) == len(ps_ref)
    for idx, (ps, mus, nm) in rec.anchors.items():
        np.testing.assert_array_equal(ps, ps_ref[idx])
        np.testing.assert_array_equal(mus, mus_ref[idx])
        if c['bb_source'] >= 0:
            nm_ref = c['model']['n_model'].reshape((-1, c['S'], rec.B))[idx, c['bb_source']]
            np.testing.assert_array_equal(nm, nm_ref)
    np.testing.assert_array_equal(rec.counts.reshape(c['counts'].shape), c['counts'])
    # the scalar half of every call (row a4): z vector, rate scale, out-of-bounds exit
    shape_names = list(lf.shape_parameters)
    for j, kw in enumerate(calls):
        kw = dict(kw)
        lt = kw.pop('livetime_days', None)
        prior, zs, scale = lf._host_terms(lt, kw)
        z_ref = c['call_z'][j]
        in_box = all(g[0] <= z <= g[-1] for g, z in zip(c['model']['anchor_z'], z_ref))
        if not in_box:
            assert prior is None
            continue
        assert prior == 0
        np.testing.assert_array_equal(zs, z_ref)
        np.testing.assert_array_equal(scale, c['raw']['call_scale'][j])
    assert shape_names == list(lf.shape_parameters)


def test_anchor_points_and_grid_order(ns):
    from blueice_amd.pdf_morphers import GridInterpolator, MORPHERS
    from blueice_amd.exceptions import NoShapeParameters
    with pytest.raises(NoShapeParameters):
        GridInterpolator({}, OrderedDict())
    sp = OrderedDict([('a', ({2: 'x', -1: 'y', 0.5: 'z'}, None, None)), ('b', ({0: 0, 1: 1}, None, None))])
    m = MORPHERS['GridInterpolator']({}, sp)
    pts = m.get_anchor_points(bounds=None)
    assert isinstance(pts, list) and isinstance(pts[0], tuple)
    assert pts == [(-1, 0), (-1, 1), (0.5, 0), (0.5, 1), (2, 0), (2, 1)]          # sorted, C order
    assert [lin for lin, _, _ in m.anchor_items()] == list(range(6))


def test_parameter_registry_and_errors(ns):
    from blueice_amd import BinnedLogLikelihood
    from blueice_amd.exceptions import (InvalidParameter, InvalidParameterSpecification, NotPreparedException)
    from blueice_amd.test_helpers import conf_for_test
    lf = BinnedLogLikelihood(conf_for_test(n_sources=2, mc=True, n_events_for_pdf=500,
                                           analysis_space=[['x', [-40, 0, 40]]]))
    lf.add_rate_parameter('s0')
    with pytest.raises(InvalidParameterSpecification):
        lf.add_shape_parameter('strlen_multiplier', {1: 'x', 2: 'hi', 3: 'wha'})      # needs base_value
    with pytest.raises(InvalidParameterSpecification):
        lf.add_shape_parameter('strlen_multiplier', ['x', 'hi'])                     # list of non-numerics
    with pytest.raises(InvalidParameterSpecification):
        lf.add_shape_parameter('some_multiplier', (0.5, 1, 2), base_value=1)         # numeric + base_value
    lf.add_shape_parameter('strlen_multiplier', {1: 'q', 2: 'hi', 3: 'wha'}, base_value=1)
    lf.add_shape_parameter('some_multiplier', (0.5, 1, 2, 4))
    assert lf.get_bounds('strlen_multiplier') == (1, 3)
    assert lf.get_bounds('some_multiplier') == (0.5, 4)
    assert lf.get_bounds() == [(1, 3), (0.5, 4)]
    assert lf.get_bounds('s0_rate_multiplier') == (0, float('inf'))
    with pytest.raises(InvalidParameter):
        lf.get_bounds('nope')
    d = np.zeros(3, dtype=[('x', float), ('source', int)])
    with pytest.raises(NotPreparedException):
        lf.set_data(d)
    with pytest.raises(NotPreparedException):
        lf()
    lf.prepare()
    assert len(lf.anchor_models) == 12 and (1, 0.5) in lf.anchor_models
    with pytest.raises(NotPreparedException):
        lf()
    lf.set_data(d)
    with pytest.raises(InvalidParameter):
        lf._host_terms(None, dict(blargh=41))
    with pytest.raises(ValueError):
        lf._host_terms(None, dict(strlen_multiplier='hi'))
    mult, settings = lf._kwargs_to_settings(s1_rate_multiplier=3)
    assert mult == [1, 3] and settings == {'strlen_multiplier': 1, 'some_multiplier': 1}
    with pytest.raises(ValueError):
        lf._host_terms(1., {})                      # no base livetime in this config -> cannot scale
    # priors are added on the host
    lf.add_rate_uncertainty('s1', 0.1)
    prior, zs, scale = lf._host_terms(None, dict(s1_rate_multiplier=1.2))
    from scipy import stats
    assert prior == stats.norm(1, 0.1).logpdf(1.2)


def test_histdd_matches_numpy_semantics():
    from blueice_amd.histdd import Histdd
    rng = np.random.default_rng(0)
    edges = [np.array([0., 1., 2.5, 7.]), np.linspace(-1, 1, 5)]
    x, y = rng.uniform(-1, 8, 500), rng.uniform(-1.2, 1.2, 500)
    x[:3] = [7., 0., 2.5]
    y[:3] = [1., -1., 0.]
    h = Histdd(x, y, bins=edges, axis_names=['x', 'y'])
    np.testing.assert_array_equal(h.histogram, np.histogramdd(np.stack([x, y], 1), bins=edges)[0])
    assert h.n == h.histogram.sum()
    np.testing.assert_allclose(h.bin_volumes(), np.outer(np.diff(edges[0]), np.diff(edges[1])))
    assert h.lookup(np.array([0.5]), np.array([-0.9]))[0] == h.histogram[0, 0]
    assert h.similar_blank_hist().histogram.sum() == 0


def test_histdd_routes_large_unweighted_batches_to_the_binning_service():
    """Inside `device_histograms(ctx)` Histdd.add hands unweighted batches of at least min_events events to
    ctx.histogram_events and keeps everything else on numpy.histogramdd; outside nothing is routed."""
    from blueice_amd.histdd import Histdd, device_histograms

    class Service:
        calls = []

        def histogram_events(self, edges, cols):
            self.calls.append(len(cols[0]))
            return np.histogramdd(np.stack(cols, 1), bins=edges)[0]

    rng = np.random.default_rng(4)
    edges = [np.linspace(0, 1, 5), np.array([0., 0.2, 1.])]
    big = [rng.random(500), rng.random(500)]
    small = [rng.random(20), rng.random(20)]
    want = Histdd(bins=edges).add(*big).add(*small).add(*big, weights=np.full(500, 2.)).histogram
    svc = Service()
    with device_histograms(svc, min_events=100):
        h = Histdd(bins=edges).add(*big).add(*small).add(*big, weights=np.full(500, 2.))
        with device_histograms(None):
            Histdd(bins=edges).add(*big)
        Histdd(bins=edges).add(*big)
    Histdd(bins=edges).add(*big)
    np.testing.assert_array_equal(h.histogram, want)
    assert svc.calls == [500, 500]
    with device_histograms(object()):                     # something that cannot bin: ignored
        np.testing.assert_array_equal(Histdd(bins=edges).add(*big).histogram, Histdd(bins=edges).add(*big).histogram)


def test_model_simulate_and_source_rates():
    from blueice_amd import Model
    from blueice_amd.test_helpers import conf_for_test, GaussianMCSource
    np.random.seed(3)
    m = Model(conf_for_test(n_sources=2, mc=True, n_events_for_pdf=2000, s1_rate_multiplier=2.))
    assert isinstance(m.sources[0], GaussianMCSource)
    np.testing.assert_allclose(m.expected_events(), [1000., 2000.])
    assert m.get_source_i('s1') == 1 and m.get_source_i(0) == 0
    with pytest.raises(ValueError):
        m.get_source_i('nope')
    d = m.simulate()
    assert 2500 < len(d) < 3500 and set(np.unique(d['source'])) == {0, 1}
    pmf, n_mc = m.pmf_grids()
    assert pmf.shape == (2, 99) and abs(pmf[0].sum() - 1) < 1e-12 and n_mc.sum() == 4000


@pytest.mark.parametrize('name', list(model_zoo.UNBINNED_CASES))
def test_unbinned_uploads_equal_reference_tensors(ns, name):
    """UnbinnedLogLikelihood.set_data must hand the device the pdf-at-events tensor the reference builds
    (likelihood.py:557-560)."""
    lf, calls, _ = model_zoo.UNBINNED_CASES[name](ns)
    c = load_case(name)
    rec = lf.ctx
    n_ev = c['bins'][0]
    assert rec.S == c['S'] and rec.B == n_ev and rec.bb_source == -1 and rec.outlier == c['outlier']
    mus_ref = c['model']['mus'].reshape((-1, c['S']))
    ps_ref = c['model']['ps'].reshape((len(mus_ref), c['S'], n_ev))
    assert len(rec.anchors) == len(ps_ref)
    for idx, (ps, mus, nm) in rec.anchors.items():
        np.testing.assert_array_equal(ps, ps_ref[idx])
        np.testing.assert_array_equal(mus, mus_ref[idx])
    for j, kw in enumerate(calls):
        kw = dict(kw)
        lt = kw.pop('livetime_days', None)
        prior, zs, scale = lf._host_terms(lt, kw)
        if prior is None:
            continue
        np.testing.assert_array_equal(zs, c['call_z'][j])
        np.testing.assert_array_equal(scale, c['raw']['call_scale'][j])


def test_source_wise_interpolation_uploads_the_expansion_of_the_per_source_tensors(ns):
    """Source-wise interpolation (likelihood.py:152-171,534-563): models are built only at the anchors some source
    needs, and the device receives the full-grid tensor whose (anchor, source) row is that source's row at ITS
    projection of the anchor -- compared with the per-source tensors the reference's interpolators hold."""
    import os
    import blueice_amd.model as model_module
    built = []
    original = model_module.Model.__init__

    def counting(self, config, **kw):
        built.append((config.get('mu'), config.get('sigma')))
        original(self, config, **kw)

    model_module.Model.__init__ = counting
    try:
        lf, calls = model_zoo.api_source_wise(ns)
    finally:
        model_module.Model.__init__ = original
    # base model + 3 anchors of mu (sigma at its base value) + 3 anchors of sigma (mu at its base value)
    assert built == [(0, 1), (-1., 1), (0., 1), (1., 1), (0, 0.8), (0, 1.), (0, 1.5)]
    assert list(lf.source_shape_parameters) == ['a', 'b'] and lf._get_shape_indices('b') == [1]
    assert lf._get_model_anchor((0.8,), 'b') == (None, 0.8)
    f = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'api_source_wise.npz'))
    rec = lf.ctx
    grid = [np.array([-1., 0., 1.]), np.array([0.8, 1., 1.5])]
    assert rec.S == 3 and rec.B == 63 and len(rec.anchors) == 9
    for lin, multi in enumerate(np.ndindex(3, 3)):
        ps, mus, _ = rec.anchors[lin]
        for i in range(3):
            dims = list(f['sw_%d_dims' % i])
            own = tuple(multi[k] for k in dims)
            np.testing.assert_array_equal(ps[i], f['sw_%d_ps' % i][own])
            assert mus[i] == f['sw_%d_mus' % i][own]
            for pos, k in enumerate(dims):
                np.testing.assert_array_equal(f['sw_%d_z_%d' % (i, pos)], grid[k])
    # the binned likelihood refuses, as the reference does
    refuse, _ = model_zoo.api_source_wise_binned(ns)
    with pytest.raises(NotImplementedError):
        refuse()


def test_reparam_and_ancillary_host_logic():
    from blueice_amd import LogAncillaryLikelihood, LogLikelihoodReParam, LogLikelihoodSum
    from blueice_amd.exceptions import InvalidParameter

    class Inner:
        rate_parameters = OrderedDict(op0=None, op1=None, bg=None)
        shape_parameters = OrderedDict(shift=({-1: -1, 1: 1}, None, None))
        pdf_base_config = dict(np0=2., np1=4., shift=0.)

        class base_model:
            config = dict(np0=2., np1=4., shift=0.)

            @staticmethod
            def simulate(rate_multipliers=None, livetime_days=None):
                return rate_multipliers, livetime_days

        def get_bounds(self, name):
            return (-1, 1) if name == 'shift' else (0, float('inf'))

        def set_data(self, d):
            self.data = d

        def __call__(self, compute_pdf=False, livetime_days=None, **kw):
            self.seen = (compute_pdf, livetime_days, dict(kw))
            return -1.5

    conv = dict(np0=((0.5, 8.), None, None), np1=((1., 6.), 'prior', 4.),
                op0_rate_multiplier=dict(params=['np0'], func=lambda a: a ** 2),
                op1_rate_multiplier=dict(params=['np0', 'np1'], func=lambda a, b: a * b))
    inner = Inner()
    lf = LogLikelihoodReParam(inner, conv)
    assert lf(np0=4., shift=0.3, bg_rate_multiplier=1.2, livetime_days=3.) == -1.5
    assert inner.seen == (False, 3., dict(op0_rate_multiplier=4., op1_rate_multiplier=2., shift=0.3, bg_rate_multiplier=1.2))
    lf()
    assert inner.seen[2] == dict(op0_rate_multiplier=1., op1_rate_multiplier=1.)
    assert list(lf.rate_parameters) == ['bg'] and list(lf.shape_parameters) == ['shift', 'np0', 'np1']
    assert lf.shape_parameters['np1'] == ({1.: 1., 6.: 6.}, 'prior', 4.)
    assert lf.get_bounds('np0') == (0.5, 8.) and lf.get_bounds('shift') == (-1, 1)
    assert lf.get_bounds() == [(-1, 1), (0.5, 8.), (1., 6.)]
    assert lf._simulate(dict(np1=8., bg=3.), livetime_days=2.) == (dict(op0=1., op1=2., bg=3.), 2.)
    lf.set_data('d')
    assert inner.data == 'd'
    names = lf.make_objective()[1]
    assert names == ['bg_rate_multiplier', 'shift', 'np0', 'np1']
    with pytest.raises(AssertionError):
        LogLikelihoodReParam(inner, dict(conv, extra=((0., 1.), None, None)))          # declared, never used
    with pytest.raises(AssertionError):
        LogLikelihoodReParam(inner, dict(np0=conv['np0'], np7=conv['np1'], op0_rate_multiplier=conv['op0_rate_multiplier'],
                                         op1_rate_multiplier=dict(params=['np0', 'np7'], func=lambda a, b: a * b)))

    anc = LogAncillaryLikelihood(lambda v, k: k * v['a'] + v['b'], ['a', 'b'], config=dict(a=2., b=10.), func_kwargs=dict(k=3.))
    assert anc() == 16. and anc(a=1.) == 13. and anc(b=0., a=0.5) == 1.5
    assert anc.get_bounds('a') == (-np.inf, np.inf) and len(anc.get_bounds()) == 2
    with pytest.raises(InvalidParameter):
        anc.get_bounds('c')
    tot = LogLikelihoodSum([lf, anc], likelihood_weights=[2., 1.])
    assert tot(np0=4., a=1.) == 2 * -1.5 + 13.
    assert inner.seen[2] == dict(op0_rate_multiplier=4., op1_rate_multiplier=2.)


def test_likelihood_sum_host_logic():
    from blueice_amd import LogLikelihoodSum
    from blueice_amd.exceptions import InvalidParameter

    class Fake:
        def __init__(self, rates, shapes, bounds, value):
            self.rate_parameters = {r: None for r in rates}
            self.shape_parameters = {k: ({b[0]: 0, b[1]: 0}, None, None) for k, b in zip(shapes, bounds)}
            self.pdf_base_config = {k: 1.0 for k in shapes}
            self._bounds, self.value, self.seen = dict(zip(shapes, bounds)), value, None

        def get_bounds(self, name):
            return self._bounds[name]

        def __call__(self, compute_pdf=False, livetime_days=None, **kw):
            self.seen = (livetime_days, dict(kw))
            return self.value

    a = Fake(['s0'], ['x', 'y'], [(-1, 2), (0, 1)], 3.0)
    b = Fake(['s1'], ['x'], [(0, 3)], 10.0)
    tot = LogLikelihoodSum([a, b], likelihood_weights=[1, 0.5])
    assert tot(x=1, y=0.5, s0_rate_multiplier=2, s1_rate_multiplier=3, livetime_days=[1., 2.]) == 3.0 + 5.0
    assert a.seen == (1., dict(x=1, y=0.5, s0_rate_multiplier=2)) and b.seen == (2., dict(x=1, s1_rate_multiplier=3))
    assert tot.get_bounds('x') == (0, 2) and tot.get_bounds('y') == (0, 1)
    assert tot.get_bounds('s0_rate_multiplier') == (0, float('inf'))
    with pytest.raises(InvalidParameter):
        tot.get_bounds('nope')
    assert tot.split_results(dict(x=1, y=2, s1_rate_multiplier=3)) == [dict(x=1, y=2), dict(x=1, s1_rate_multiplier=3)]
    objective, names, guess, bounds = tot.make_objective()
    assert names == ['s0_rate_multiplier', 's1_rate_multiplier', 'x', 'y']


@pytest.mark.parametrize('name,bb', [('mini3', -1), ('mini4bb', 0)])
def test_synthetic_model_through_the_plugin_route(monkeypatch, name, bb):
    """SyntheticModel.likelihood() builds the model from Source plug-ins (TemplateSource) through the ordinary
    config -> Model -> prepare() route: what reaches the device is exactly the synthetic tensor, anchor by anchor."""
    import blueice_amd.likelihood as lk
    from blueice_amd.synthetic import SyntheticModel, TemplateSource
    monkeypatch.setattr(lk, 'DeviceContext', RecordingContext)
    RecordingContext.instances.clear()
    m = SyntheticModel.named(name, bb_source=bb)
    lf = m.likelihood()
    rec = RecordingContext.instances[-1]
    assert rec.S == m.S and rec.B == m.B and rec.bb_source == bb and len(rec.anchors) == m.A
    for g, want in zip(rec.anchor_z, m.anchor_z):
        np.testing.assert_array_equal(g, want)
    for a in range(m.A):
        ps, mus, nm = rec.anchors[a]
        np.testing.assert_array_equal(ps, m.anchor_ps(a))
        np.testing.assert_array_equal(mus, m.anchor_mus(a))
        if bb >= 0:
            np.testing.assert_array_equal(nm, m.anchor_n_model(a))
    assert list(lf.shape_parameters) == ['shape%d' % i for i in range(m.d)]
    assert all(isinstance(s, TemplateSource) for s in lf.base_model.sources)
    assert lf.get_bounds('shape0') == (float(m.anchor_z[0][0]), float(m.anchor_z[0][-1]))
    with pytest.raises(ValueError):                       # a TemplateSource exists on the anchors only
        lf._compute_single_model(shape0=0.123)


def test_prepare_with_a_thread_pool_builds_the_same_anchor_models(ns):
    """prepare(n_cores=4): anchor models built by a thread pool, same tensors in the same anchor order."""
    lf, _, _ = model_zoo.CASES['d2_nonuniform'](ns)
    first = dict(RecordingContext.instances[-1].anchors)
    order = list(lf.anchor_models)
    lf.prepare(n_cores=4)
    again = RecordingContext.instances[-1].anchors
    assert list(lf.anchor_models) == order and sorted(again) == sorted(first) and len(first) > 1
    for k in first:
        np.testing.assert_array_equal(again[k][0], first[k][0])
        np.testing.assert_array_equal(again[k][1], first[k][1])
