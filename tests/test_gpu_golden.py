"""GPU parity proper: libblueice_hip (through the C ABI) against the golden vectors produced by
the reference and against the CPU oracle on the same inputs.

Tolerance (BASELINE.json north_star): |gpu - cpu| <= 1e-10 * max(1, |cpu|); +-inf / nan exact."""
import numpy as np
import pytest

from golden_util import case_names, load_case, rate_scale_of, same

pytestmark = pytest.mark.gpu

RTOL = 1e-10


@pytest.fixture(scope='module')
def ctx():
    from blueice_amd.device import DeviceContext
    c = DeviceContext(0)
    yield c
    c.close()


def upload_case(ctx, c):
    bb = c['bb_source']
    ctx.upload_model(c['model']['anchor_z'], c['model']['ps'], c['model']['mus'],
                     n_model=c['model']['n_model'] if bb >= 0 else None, bb_source=bb)
    if c.get('allow_negative') is not None:
        ctx.set_allow_negative([1 if a else 0 for a in c['allow_negative']])
    ctx.upload_counts(c['counts'])


@pytest.mark.parametrize('sparse', [0, 2])
@pytest.mark.parametrize('name', case_names())
def test_eval_matches_reference_goldens(ctx, name, sparse):
    """sparse=0: every bin is visited (the dense morph+reduce kernel).  sparse=2: whenever it is exact
    (non-negative templates, no Beeston-Barlow) only the non-empty bins are visited, the empty ones enter
    through precomputed row sums -- same numbers required."""
    from blueice_amd import _capi
    c = load_case(name)
    ctx.set_param('sparse', sparse)
    upload_case(ctx, c)
    if sparse:
        assert ctx.get_param('csr_ready') == 1
        assert ctx.get_param('compact_ready') == (1 if c['bb_source'] < 0 else 0)
    uses_sparse = bool(sparse) and c['bb_source'] < 0 and not any(c['allow_negative'] or [])
    n = len(c['call_ll'])
    rs = np.array([rate_scale_of(c, j) for j in range(n)])
    # one by one (the lf(**kw) form) ...
    single = []
    for j in range(n):
        ll, st = ctx.eval(c['call_z'][j], rs[j])
        single.append((ll[0], st[0]))
    # ... and as one batch (the scan form); both must agree with the reference
    batch, bst = ctx.eval(c['call_z'], rs)
    for j in range(n):
        asserts = ('call_asserts_%d' % j) in c['raw'].files
        for ll, st in (single[j], (batch[j], bst[j])):
            if asserts:
                assert st & (_capi.ST_BB_ROOT1 | _capi.ST_BB_NEG), (name, j, ll, st)
                continue
            assert same(ll, c['call_ll'][j], RTOL), (name, j, ll, c['call_ll'][j])
        assert same(single[j][0], batch[j], 1e-13) or asserts
    ctx.set_param('sparse', 1)


@pytest.mark.parametrize('name', case_names())
def test_interpolators_and_full_output(ctx, name):
    """Compatibility mode: the morpher closures and full_output=True."""
    from oracle import blueice_oracle as orc
    c = load_case(name)
    upload_case(ctx, c)
    f = c['raw']
    B = int(np.prod(c['bins']))
    for j in range(min(len(c['call_ll']), 6)):
        z = c['call_z'][j]
        if not orc.in_bounds(c['model']['anchor_z'], z):
            with pytest.raises(ValueError):
                ctx.interpolate('ps', z)
            continue
        # bit-identical: same corner order, separate multiply and add
        np.testing.assert_array_equal(ctx.interpolate('ps', z),
                                      orc.interpolate(c['model']['anchor_z'], c['model']['ps'], z).reshape(c['S'], B))
        np.testing.assert_array_equal(ctx.interpolate('mus', z),
                                      orc.interpolate(c['model']['anchor_z'], c['model']['mus'], z))
        if c['bb_source'] >= 0:
            nm = orc.interpolate(c['model']['anchor_z'], c['model']['n_model'], z)[c['bb_source']]
            np.testing.assert_array_equal(ctx.interpolate('n_model', z), nm.reshape(B))
    for key in f.files:
        if key.startswith('full_') and key.endswith('_mus'):
            j = int(key.split('_')[1])
            ll, mus, ps, st = ctx.eval_full(c['call_z'][j], rate_scale_of(c, j))
            assert same(ll, c['call_ll'][j], RTOL)
            np.testing.assert_allclose(mus, f['full_%d_mus' % j], rtol=1e-12, atol=0)
            np.testing.assert_allclose(ps, f['full_%d_ps' % j].reshape(c['S'], B), rtol=1e-12, atol=1e-300)


def test_status_bits(ctx):
    from blueice_amd import _capi
    c = load_case('d3_small')
    upload_case(ctx, c)
    ll, st = ctx.eval(c['call_z'][-1], rate_scale_of(c, len(c['call_ll']) - 1))     # nan z
    assert ll[0] == -np.inf and st[0] == _capi.ST_OUT_OF_BOUNDS
    ll, st = ctx.eval([0., 0., 0.], [1., -1., 1., 1.])
    assert ll[0] == -np.inf and st[0] == _capi.ST_UNPHYSICAL
    ll, st = ctx.eval([0., 0., 0.], None, dataset=[3])
    assert st[0] == _capi.ST_BAD_DATASET


def test_state_errors():
    from blueice_amd.device import DeviceContext
    from blueice_amd.exceptions import NotPreparedException
    c = DeviceContext(0)
    with pytest.raises(NotPreparedException):
        c.d, c.S, c.B = 0, 1, 1
        c.eval(None, [[1.]])
    c.close()


def test_device_histogram_equals_numpy_histogramdd(ctx):
    """bi_upload_events (set_data on the device) against numpy.histogramdd, the semantics of the reference's
    binning (likelihood.py:608-609): non-uniform edges, events on edges, out of range, nan."""
    rng = np.random.default_rng(5)
    edges = [np.array([0., 1., 2.5, 7.]), np.linspace(-1, 1, 6), np.array([-3., 0., 0.5, 4., 4.25])]
    shape = tuple(len(e) - 1 for e in edges)
    B = int(np.prod(shape))
    ctx.upload_model([], np.full((1, B), 1.0 / B), np.array([10.]))
    ctx.set_analysis_space(edges)
    N = 20000
    x = rng.uniform(-1, 8, N)
    y = rng.uniform(-1.3, 1.3, N)
    w = rng.uniform(-4, 5, N)
    x[:6] = [7., 0., 2.5, 1., 7.0000001, np.nan]
    y[:6] = [1., -1., 0.2, -0.6, 0., 0.]
    w[:6] = [4.25, -3., 0.5, 0., 1., 1.]
    ctx.upload_events(x, y, w)
    got = ctx.download_counts(0).reshape(shape)
    keep = ~(np.isnan(x) | np.isnan(y) | np.isnan(w))
    want = np.histogramdd(np.stack([x[keep], y[keep], w[keep]], 1), bins=edges)[0]
    np.testing.assert_array_equal(got, want)
    assert got.sum() < N
    ll, st = ctx.eval(None, [[1.]])
    assert np.isfinite(ll[0]) and st[0] == 0
    ctx.upload_events(np.zeros(0), np.zeros(0), np.zeros(0))            # empty dataset
    assert ctx.download_counts(0).sum() == 0
    with pytest.raises(ValueError):
        ctx.set_analysis_space([np.array([0., 1.])])                    # bin count mismatch


def test_template_histograms_on_the_device():
    """bi_histogram_events -- the binning service behind Histdd.add while anchor models are built -- against
    numpy.histogramdd (what multihist's add applies, blueice/source.py:287-299): model-free context, 1 to 4 axes,
    non-uniform edges, events on inner and outer edges, out of range, nan and inf, no events at all."""
    from blueice_amd.device import DeviceContext
    from blueice_amd.histdd import Histdd, device_histograms
    rng = np.random.default_rng(8)
    c = DeviceContext(0)
    try:
        for edges in ([np.linspace(-3, 3, 41)],
                      [np.array([0., 1., 2.5, 7.]), np.linspace(-1, 1, 6)],
                      [np.array([0., 1., 2.5, 7.]), np.linspace(-1, 1, 6), np.array([-3., 0., 0.5, 4., 4.25])],
                      [np.linspace(0, 1, 8), np.linspace(0, 2, 5), np.array([0., 0.1, 1.]), np.linspace(-1, 0, 4)]):
            N = 50000
            cols = [rng.uniform(e[0] - 0.3, e[-1] + 0.3, N) for e in edges]
            for k, e in enumerate(edges):                       # exact edge values, first and last included
                cols[k][:len(e)] = e
            cols[0][len(edges[0]) + 1] = np.nan
            cols[-1][len(edges[-1]) + 2] = np.inf
            got = c.histogram_events(edges, cols)
            keep = np.all([np.isfinite(x) for x in cols], axis=0)
            want = np.histogramdd(np.stack([x[keep] for x in cols], 1), bins=edges)[0]
            np.testing.assert_array_equal(got, want)
            assert got.shape == want.shape and 0 < got.sum() < N
            np.testing.assert_array_equal(c.histogram_events(edges, [np.zeros(0)] * len(edges)), np.zeros(want.shape))
            # through Histdd: large batches go to the device, small ones and weighted ones stay with numpy
            host = Histdd(bins=edges).add(*[x[keep] for x in cols])
            with device_histograms(c, min_events=1000):
                dev = Histdd(bins=edges).add(*cols).add(*[x[:10] for x in cols])
                weighted = Histdd(bins=edges).add(*[x[keep] for x in cols], weights=np.full(int(keep.sum()), 0.5))
            host.add(*[x[:10] for x in cols])
            np.testing.assert_array_equal(dev.histogram, host.histogram)
            np.testing.assert_array_equal(weighted.histogram * 2, want)
        with pytest.raises(ValueError):
            c.histogram_events([np.array([0., 1., 1.])], [np.zeros(3)])     # edges not strictly ascending
    finally:
        c.close()


def test_prepare_builds_the_same_templates_with_device_histograms():
    """prepare() of a likelihood whose sources estimate their densities from 10^5 Monte Carlo events each: templates,
    MC counts and rates uploaded with the device binning the samples equal those of the numpy.histogramdd route
    bit for bit -- and so does the likelihood."""
    import model_zoo
    ns = model_zoo.namespace_of('blueice_amd')
    values = []
    for on in (True, False):
        np.random.seed(12)
        conf = ns.conf_for_test(n_sources=2, mc=True, n_events_for_pdf=100000, events_per_day=50.,
                                analysis_space=[['x', np.linspace(-5, 5, 201)]])
        lf = ns.BinnedLogLikelihood(conf, likelihood_config=dict(device_histograms=on))
        lf.add_rate_parameter('s0')
        lf.add_shape_parameter('mu', (-1., 0., 1.))
        lf.prepare()
        d = np.zeros(400, dtype=[('x', float), ('source', int)])
        d['x'] = np.random.default_rng(3).normal(0.2, 1.1, 400)
        lf.set_data(d)
        values.append((lf(mu=0.3, s0_rate_multiplier=1.2), lf.ps_interpolator(np.array([0.3])),
                       lf.anchor_models[(1.0,)].sources[1]._n_events_histogram.histogram))
    assert values[0][0] == values[1][0] and np.isfinite(values[0][0])
    np.testing.assert_array_equal(values[0][1], values[1][1])
    np.testing.assert_array_equal(values[0][2], values[1][2])
    assert values[0][2].sum() > 90000


def test_device_log_accuracy(ctx):
    """The lean logarithm of the per-bin terms against numpy (glibc) over the whole double range:
    <= 1 ulp on normal arguments, library behaviour on 0 / denormal / negative / inf / nan."""
    from blueice_amd._capi import ptr
    rng = np.random.default_rng(0)
    x = np.concatenate([10.0 ** rng.uniform(-307, 308, 200000), rng.uniform(0.5, 2.0, 200000),
                        10.0 ** rng.uniform(-323, -307, 20000),        # denormals
                        rng.uniform(0.98, 1.02, 100000),               # where log changes sign
                        2.0 ** rng.integers(-1000, 1000, 2000) * np.repeat([0.6875, 1.375], 1000),   # table interval edges
                        1.0 + rng.uniform(-1e-6, 1e-6, 50000), np.array([1.0, 2.0, 0.5, np.e, 0.70710678118654752, 1e-320,
                        0.0, -1.0, np.inf, np.nan, 2.2250738585072014e-308, 1.7976931348623157e308])])
    out = np.empty_like(x)
    assert ctx._lib.bi_selftest_log(ctx._h, len(x), ptr(x), ptr(out)) == 0
    with np.errstate(all='ignore'):
        ref = np.log(x)
    fin = np.isfinite(ref)
    ulp = np.abs(out[fin] - ref[fin]) / np.spacing(np.abs(ref[fin]))
    assert ulp.max() <= 1.0, ulp.max()
    assert np.mean(ulp == 0) > 0.85
    assert out[-6] == -np.inf and np.isnan(out[-5]) and out[-4] == np.inf and np.isnan(out[-3])
    assert out[len(x) - 12] == 0.0            # log(1) exactly


def test_infinite_rate_follows_the_reference():
    """A rate of +inf is legal next to a source that may go negative (blueice/likelihood.py:403-415).  The reference
    scales the INTERPOLATED template, inf * p(z)_b: +inf where p(z)_b > 0 (then -inf without data, nan with), nan where
    p(z)_b == 0.  Corner templates with exact zeros in DIFFERENT bins must not change that: single calls and batches,
    empty and non-empty data, on and off anchors."""
    from oracle import blueice_oracle as orc
    from blueice_amd.device import DeviceContext
    rng = np.random.default_rng(3)
    anchor_z = [np.array([-1., 0., 2.]), np.array([0., 1.])]
    S, B = 3, 40
    ps = rng.random((3, 2, S, B)) + 0.01
    ps[0, 0, 0, 3] = 0.0                        # a zero in one corner template only: the interpolated value stays > 0 off that anchor
    ps[:, :, 0, 7] = 0.0                        # a zero at every anchor: p(z)_7 == 0 everywhere
    ps /= ps.sum(axis=-1, keepdims=True)
    mus = rng.uniform(5, 50, (3, 2, S))
    model = dict(anchor_z=anchor_z, ps=ps, mus=mus, n_model=None)
    allow = [False, True, False]
    z = np.array([[-1., 0.], [-0.5, 0.3], [0., 1.], [1.1, 0.7], [2., 0.], [-1., 0.4]])
    ctx = DeviceContext(0)
    ctx.upload_model(anchor_z, ps, mus)
    ctx.set_allow_negative([1 if a else 0 for a in allow])
    for src_inf in (0, 2):                      # source 0 has the zeros, source 2 has none
        r = np.ones((len(z), S))
        r[:, src_inf] = np.inf
        r[::2, 1] = -0.3
        for counts in (np.zeros(B), rng.poisson(2.0, B).astype(float)):
            want = np.array([orc.loglikelihood(model, counts, z[i], r[i], allow_negative=allow) for i in range(len(z))])
            for sparse in (0, 1):
                ctx.set_param('sparse', sparse)
                ctx.upload_counts(counts)
                batch, st = ctx.eval(z, r)
                assert not st.any()
                for i in range(len(z)):
                    one, st1 = ctx.eval(z[i], r[i])
                    for got in (one[0], batch[i]):
                        assert (np.isnan(got) and np.isnan(want[i])) or got == want[i], (src_inf, counts.sum(), sparse, i, got, want[i])
            if src_inf == 2 and counts.sum() == 0:
                assert np.all(want == -np.inf)              # +inf expected everywhere, nothing seen
            if src_inf == 0:
                assert np.all(np.isnan(want))               # bin 7: inf * 0
    ctx.close()
