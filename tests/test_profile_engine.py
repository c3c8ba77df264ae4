"""The batched profile-fit engine on the CPU: the optimiser itself (blueice_amd.profile.batched_minimize) on functions
with known minima, and the whole of `bestfit_batched` / the batched scan and interval drivers on a likelihood stand-in
served by the oracle (tests/oracle_lf.py) -- against scipy's own minimiser run problem by problem, which is what the
reference does (blueice/inference.py:131-178,332-443)."""
import numpy as np
import pytest
from scipy.optimize import minimize

from golden_util import load_case
from oracle_lf import OracleLikelihood


def test_batched_minimize_quadratics_with_bounds():
    from blueice_amd.profile import batched_minimize
    rng = np.random.default_rng(0)
    P, F = 200, 4
    A = rng.normal(size=(P, F, F))
    A = np.einsum('pij,pkj->pik', A, A) + 0.5 * np.eye(F)
    c = rng.normal(size=(P, F)) * 3
    lo, hi = np.array([-1., -np.inf, 0., -2.]), np.array([1., np.inf, np.inf, 2.])

    def fun(x, rows):
        dx = x - c[rows]
        Ad = np.einsum('pij,pj->pi', A[rows], dx)
        return 0.5 * np.sum(dx * Ad, axis=1), Ad

    x, f, info = batched_minimize(fun, np.zeros((P, F)), lo, hi, gtol=1e-7)
    assert (info['converged'] | info['stalled']).all() and info['stalled'].sum() <= 2      # (stalled: at the rounding floor)
    for p in range(0, P, 17):
        ref = minimize(lambda v: fun(v[None], np.array([p]))[0][0], np.zeros(F), jac=lambda v: fun(v[None], np.array([p]))[1][0],
                       bounds=list(zip(lo, hi)), method='L-BFGS-B', options=dict(ftol=1e-15, gtol=1e-10))
        assert f[p] <= ref.fun + 1e-9 * max(1, abs(ref.fun))
        np.testing.assert_allclose(x[p], ref.x, atol=2e-5)


def test_batched_minimize_rosenbrock_and_dead_starts():
    from blueice_amd.profile import batched_minimize
    a = np.linspace(0.5, 2.0, 32)

    def fun(x, rows):
        aa = a[rows]
        f = (aa - x[:, 0]) ** 2 + 20.0 * (x[:, 1] - x[:, 0] ** 2) ** 2
        g = np.stack([-2 * (aa - x[:, 0]) - 80.0 * x[:, 0] * (x[:, 1] - x[:, 0] ** 2), 40.0 * (x[:, 1] - x[:, 0] ** 2)], axis=1)
        bad = rows == 5                                   # a problem whose objective is not finite anywhere
        return np.where(bad, np.inf, f), g

    x, f, info = batched_minimize(fun, np.tile([-1.0, 1.0], (32, 1)), np.full(2, -np.inf), np.full(2, np.inf), gtol=1e-7, max_iter=500)
    ok = np.arange(32) != 5
    assert info['failed'][5] and info['converged'][ok].all()
    np.testing.assert_allclose(x[ok, 0], a[ok], atol=1e-5)
    np.testing.assert_allclose(x[ok, 1], a[ok] ** 2, atol=1e-5)


@pytest.fixture(scope='module')
def d2():
    case = load_case('d2_nonuniform')
    return case


@pytest.mark.parametrize('analytic', [False, True])
def test_profiled_scan_equals_per_point_scipy_fits(d2, analytic):
    """24 hypotheses of s0's rate, everything else (2 rates, 2 shape parameters) floating: the batched engine against
    scipy run hypothesis by hypothesis on the same function."""
    from blueice_amd.profile import bestfit_batched
    lf = OracleLikelihood(d2['model'], d2['counts'], ['shift', 'stretch'], analytic=analytic)
    grid = np.linspace(0.3, 2.5, 24)
    best, ll, info = bestfit_batched(lf, points={'s0_rate_multiplier': grid}, return_info=True)
    assert list(best) == ['s1_rate_multiplier', 's2_rate_multiplier', 'shift', 'stretch']
    assert info['analytic_gradient'] == analytic
    assert (info['converged'] | info['stalled']).all()
    bounds = [(0, None), (0, None), lf.get_bounds('shift'), lf.get_bounds('stretch')]
    for j in range(0, 24, 5):
        f = lambda v: -lf(s0_rate_multiplier=grid[j], s1_rate_multiplier=v[0], s2_rate_multiplier=v[1], shift=v[2], stretch=v[3])
        ref = min((minimize(f, x0, bounds=bounds, method='L-BFGS-B', options=dict(ftol=1e-14, gtol=1e-8))
                   for x0 in ([1, 1, 0, 0.5], [best[k][j] for k in best])), key=lambda r: r.fun)
        assert ll[j] >= -ref.fun - 1e-6 * abs(ref.fun), (j, ll[j], -ref.fun)
        # and it is a value the likelihood really takes at the returned parameters
        assert abs(lf(s0_rate_multiplier=grid[j], **{k: best[k][j] for k in best}) - ll[j]) <= 1e-9 * abs(ll[j])
    # a batched engine pays per ITERATION, not per problem: far fewer device calls than fits x iterations
    assert info["calls"] < 800, info["calls"]


def test_fixed_scalars_guesses_and_nothing_to_fit(d2):
    from blueice_amd.exceptions import NoOpimizationNecessary
    from blueice_amd.profile import bestfit_batched
    lf = OracleLikelihood(d2['model'], d2['counts'], ['shift', 'stretch'])
    best, ll = bestfit_batched(lf, points={'shift': np.array([-0.5, 0.2, 1.0])}, stretch=1.0, s2_rate_multiplier=1.0,
                               guess={'s0_rate_multiplier': np.array([1.0, 1.2, 0.8])})
    assert list(best) == ['s0_rate_multiplier', 's1_rate_multiplier'] and ll.shape == (3,)
    one, ll1 = bestfit_batched(lf, shift=0.2, stretch=1.0, s2_rate_multiplier=1.0)          # P = 1: a plain best fit
    assert abs(ll1[0] - ll[1]) <= 1e-7 * abs(ll[1])
    with pytest.raises(NoOpimizationNecessary):
        bestfit_batched(lf, points={'shift': np.array([0.1])}, stretch=1.0, s0_rate_multiplier=1., s1_rate_multiplier=1., s2_rate_multiplier=1.)


def test_interval_and_profiled_scan_drivers_on_the_engine(d2):
    """inference.one_parameter_interval / likelihood_ratio_scan on the batched engine against the reference's own loop
    structure (brentq over nested per-hypothesis fits; a Python loop of fits) run with scipy on the same function."""
    from scipy import stats
    from scipy.optimize import brentq
    from blueice_amd import inference
    lf = OracleLikelihood(d2['model'], d2['counts'], ['shift', 'stretch'], analytic=True)
    fixed = dict(stretch=1.0, s2_rate_multiplier=1.0)
    bounds = [(0, None), lf.get_bounds('shift')]

    def fit_seq(**kw):                         # profile over s1's rate and the shift, sequentially, as the reference would
        f = lambda v: -lf(s1_rate_multiplier=v[0], shift=v[1], **kw)
        return -min(minimize(f, x0, bounds=bounds, method='L-BFGS-B', options=dict(ftol=1e-15, gtol=1e-9)).fun
                    for x0 in ([1.0, 0.0], [0.5, 0.3]))

    f_all = lambda v: -lf(s0_rate_multiplier=v[0], s1_rate_multiplier=v[1], shift=v[2], **fixed)
    glob = minimize(f_all, [1, 1, 0], bounds=[(0, None)] + bounds, method='L-BFGS-B', options=dict(ftol=1e-15, gtol=1e-9))
    crit = stats.norm.ppf(0.9) ** 2
    want = brentq(lambda h: 2 * (-glob.fun - fit_seq(s0_rate_multiplier=h, **fixed)) - crit, glob.x[0], 6.0, xtol=1e-12)
    before = lf.n_batches
    got = inference.one_parameter_interval(lf, 's0_rate_multiplier', bound=6.0, kind='upper', confidence_level=0.9, **fixed)
    assert abs(got - want) <= 1e-6 * want, (got, want)
    assert lf.n_batches - before < 1500          # rounds of batched fits, not thousands of nested scalar calls
    t_seq = lambda h, q: 2 * (-glob.fun - fit_seq(s0_rate_multiplier=h, **fixed)) - stats.norm.ppf(q) ** 2
    if t_seq(0.0, 0.4) > 0:                     # (sources that resemble each other: a vanishing s0 is only mildly disfavoured)
        lo_hi = inference.one_parameter_interval(lf, 's0_rate_multiplier', bound=(0.0, 6.0), kind='central', confidence_level=0.2, **fixed)
        assert abs(lo_hi[0] - brentq(t_seq, 0.0, glob.x[0], args=(0.4,), xtol=1e-12)) <= 1e-6
        assert abs(lo_hi[1] - brentq(t_seq, glob.x[0], 6.0, args=(0.6,), xtol=1e-12)) <= 1e-6
    else:
        with pytest.raises(ValueError, match='different signs'):
            inference.one_parameter_interval(lf, 's0_rate_multiplier', bound=(0.0, 6.0), kind='central', confidence_level=0.2, **fixed)
    with pytest.raises(ValueError, match='different signs'):
        inference.one_parameter_interval(lf, 's0_rate_multiplier', bound=glob.x[0] * 1.001, kind='upper', **fixed)
    grid = np.linspace(0.4, 2.0, 9)
    scan = inference.likelihood_ratio_scan(lf, ('s0_rate_multiplier', grid), **fixed)
    want_scan = np.array([fit_seq(s0_rate_multiplier=h, **fixed) for h in grid])
    np.testing.assert_allclose(scan, want_scan.max() - want_scan, atol=2e-6)


def test_fit_routine_options_pass_through_the_drivers(d2):
    """The reference's drivers forward their **kwargs to the fit routine (blueice/inference.py:332-443), so callers hand
    bestfit_scipy's options there: those calls must reach bestfit_scipy -- not the engine's list of fixed parameters, where a
    strict likelihood raises InvalidParameter."""
    from blueice_amd import inference
    from blueice_amd.exceptions import InvalidParameter
    lf = OracleLikelihood(d2['model'], d2['counts'], ['shift', 'stretch'], analytic=False)
    with pytest.raises(InvalidParameter):
        lf(pass_bounds_to_minimizer=True)                                  # the stand-in is as strict as the device class
    fixed = dict(stretch=1.0, shift=0.1, s2_rate_multiplier=1.0)
    plain = inference.one_parameter_interval(lf, 's0_rate_multiplier', bound=6.0, kind='upper', confidence_level=0.9, **fixed)
    with_opts = inference.one_parameter_interval(lf, 's0_rate_multiplier', bound=6.0, kind='upper', confidence_level=0.9,
                                                 pass_bounds_to_minimizer=True, minimize_kwargs=dict(tol=1e-9), **fixed)
    assert abs(with_opts - plain) <= 2e-3 * plain
    grid = np.linspace(0.6, 1.6, 5)
    a = inference.likelihood_ratio_scan(lf, ('s0_rate_multiplier', grid), **fixed)
    b = inference.likelihood_ratio_scan(lf, ('s0_rate_multiplier', grid), pass_bounds_to_minimizer=True, **fixed)
    np.testing.assert_allclose(a, b, atol=2e-3)
    # nothing left to fit: the options are dropped, the grid is one batched call
    c = inference.likelihood_ratio_scan(lf, ('s0_rate_multiplier', grid), s1_rate_multiplier=1.0, rates_in_log_space=True, **fixed)
    assert c.shape == (5,) and c.min() == 0


def test_stencil_batching_gives_scipy_its_own_differences(d2):
    """bestfit_scipy with the finite-difference stencil evaluated as one batch: same minimiser, same differences, a
    fraction of the calls -- the fit it returns is the fit of the scalar stream."""
    from scipy.optimize._numdiff import approx_derivative
    from blueice_amd import inference
    lf = OracleLikelihood(d2['model'], d2['counts'], ['shift', 'stretch'])
    for name in inference.__all__:
        setattr(OracleLikelihood, name, getattr(inference, name))
    fixed = dict(s2_rate_multiplier=1.0, stretch=1.0)
    f_st, names, guess, _ = lf.make_objective(with_gradient='stencil', **fixed)
    f_sc, names2, _, _ = lf.make_objective(**fixed)
    assert names == names2 == ['s0_rate_multiplier', 's1_rate_multiplier', 'shift']
    x = np.array([1.2, 0.7, -0.3])
    v, g = f_st(x)
    assert v == f_sc(x)
    np.testing.assert_allclose(g, approx_derivative(f_sc, x, method='2-point'), rtol=1e-9, atol=1e-9)     # scipy's own numbers
    # scipy's default minimiser on both forms: same path (same differences), a third of the device calls
    lf.n_calls = lf.n_batches = 0
    r_b = minimize(f_st, guess, jac=True)
    batches = lf.n_batches
    lf.n_calls = lf.n_batches = 0
    r_s = minimize(f_sc, guess)
    scalar_calls = lf.n_batches
    assert r_b.nit == r_s.nit and abs(r_b.fun - r_s.fun) <= 1e-7 * abs(r_s.fun)
    assert batches < scalar_calls / 2.5, (batches, scalar_calls)
    # ... and through bestfit_scipy (including its Nelder-Mead fallback, which takes no gradient): the same fit
    best_b, ll_b = lf.bestfit_scipy(**fixed)
    best_s, ll_s = lf.bestfit_scipy(batch_stencil=False, **fixed)
    assert abs(ll_b - ll_s) <= 1e-6 * abs(ll_s)
    np.testing.assert_allclose(list(best_b.values()), list(best_s.values()), rtol=2e-3, atol=2e-3)
    # with bounds handed to the minimiser the stencil turns around at an upper bound, as scipy's does
    f_b, _, _, bounds = lf.make_objective(with_gradient='stencil', stencil_respects_bounds=True, **fixed)
    top = np.array([1.0, 1.0, lf.get_bounds('shift')[1]])
    v, g = f_b(top)
    assert np.all(np.isfinite(g))
    assert not np.all(np.isfinite(f_st(top)[1]))              # unbounded: the forward step leaves the anchor box (+inf), as in scipy


def test_first_crossing_finds_brentqs_root():
    """The batched bracket search of one_parameter_interval on plain functions: the root nearest to the starting end, to
    brentq's precision, in a handful of batched rounds; brentq's error on equal signs; exact zeros at the ends."""
    from scipy.optimize import brentq
    from blueice_amd.inference import _first_crossing
    calls = []

    def batched(f):
        def tfun(hs):
            calls.append(len(hs))
            return np.array([f(h) for h in hs])
        return tfun

    for f, a, b in ((lambda x: x ** 3 - 2 * x - 5, 2.0, 3.0), (lambda x: np.exp(-x) - 0.1, 0.0, 50.0),
                    (lambda x: 2.7 - (x - 1) ** 2, 1.0, 40.0), (lambda x: np.cos(x), 3.0, 0.0)):
        calls.clear()
        got = _first_crossing(batched(f), a, b)
        want = brentq(f, min(a, b), max(a, b), xtol=1e-13)
        assert abs(got - want) <= 1e-9 * max(1.0, abs(want)), (got, want)
        assert len(calls) <= 9 and sum(calls) <= 140           # rounds, evaluations
    # several roots: the one nearest to a
    f = lambda x: np.sin(x)
    assert abs(_first_crossing(batched(f), 2.0, 11.0) - np.pi) <= 1e-9
    assert abs(_first_crossing(batched(f), 11.0, 2.0) - 3 * np.pi) <= 1e-9
    with pytest.raises(ValueError, match='different signs'):
        _first_crossing(batched(lambda x: x * x + 1), -1.0, 1.0)
    assert _first_crossing(batched(lambda x: x - 1.0), 1.0, 5.0) == 1.0
    assert _first_crossing(batched(lambda x: x - 5.0), 1.0, 5.0) == 5.0


def test_batched_minimize_starting_on_kinks():
    """A variable that sits ON a kink of the function (a shape parameter at an anchor: where every fit starts) has two
    one-sided slopes.  `fun` reports the right one, as the device does (the point belongs to the upper cell); the
    optimiser asks for the left one with one more row just below, holds the variable when both point uphill (and can
    then call the point converged), and otherwise leaves the kink down the steeper side."""
    from blueice_amd.profile import batched_minimize
    # per problem: f = w * |x0| + (x0 - m)^2 + (x1 - 1)^2, kink of x0 at 0; minimum at x0 = 0 iff |2 m| <= w
    m = np.array([0.0, 0.04, -0.04, 0.3, -0.3, 0.3, -0.3])
    w = np.array([0.1, 0.1, 0.1, 0.1, 0.1, 1.0, 1.0])

    def fun(x, rows):
        x0, x1 = x[:, 0], x[:, 1]
        sign = np.where(x0 >= 0, 1.0, -1.0)                       # at 0: the right slope
        f = w[rows] * np.abs(x0) + (x0 - m[rows]) ** 2 + (x1 - 1) ** 2
        return f, np.stack([w[rows] * sign + 2 * (x0 - m[rows]), 2 * (x1 - 1)], axis=1)

    lo, hi = np.array([-2., -5.]), np.array([2., 5.])
    x, f, info = batched_minimize(fun, np.zeros((7, 2)), lo, hi, gtol=1e-8, kinks=[np.array([-1., 0., 1.]), np.zeros(0)])
    want = np.where(np.abs(2 * m) <= w, 0.0, m - np.sign(m) * w / 2)
    np.testing.assert_allclose(x[:, 0], want, atol=1e-6)
    np.testing.assert_allclose(x[:, 1], 1.0, atol=1e-6)
    assert info['converged'].all() and info['kink_calls'] > 0
    assert np.all(x[np.abs(2 * m) <= w, 0] == 0.0)                # held exactly on the kink, not near it
    # without the kinks the one-sided slope at the start looks like a descent direction that is none
    x2, f2, info2 = batched_minimize(fun, np.zeros((7, 2)), lo, hi, gtol=1e-8)
    assert not info2['converged'][:3].all() and np.all(f <= f2 + 1e-12)


# ---- the C++ loop (csrc/bi_fit.h) against the numpy form it was ported from ---------------------------------------------
def _engine_cases():
    rng = np.random.default_rng(3)
    P, F = 60, 3
    A = rng.normal(size=(P, F, F))
    A = np.einsum('pij,pkj->pik', A, A) + 0.3 * np.eye(F)
    c = rng.normal(size=(P, F)) * 2

    def quad(x, rows):
        dx = x - c[rows]
        Ad = np.einsum('pij,pj->pi', A[rows], dx)
        return 0.5 * np.sum(dx * Ad, axis=1), Ad

    w = np.linspace(0.2, 3.0, P)
    m = np.linspace(-1.5, 1.5, P)

    def vee(x, rows):                                    # |x0| kinks at -1, 0, 1 scaled differently per problem
        x0, x1, x2 = x[:, 0], x[:, 1], x[:, 2]
        sign = np.where(x0 >= 0, 1.0, -1.0)
        f = w[rows] * np.abs(x0) + (x0 - m[rows]) ** 2 + (x1 - 1) ** 2 + 0.5 * (x2 + 0.3 * x1) ** 2
        g = np.stack([w[rows] * sign + 2 * (x0 - m[rows]), 2 * (x1 - 1) + 0.3 * (x2 + 0.3 * x1), x2 + 0.3 * x1], axis=1)
        dead = rows == 7
        return np.where(dead, np.nan, f), g

    lo, hi = np.array([-2., -np.inf, 0.]), np.array([2., np.inf, np.inf])
    return [('quadratics with bounds', quad, np.zeros((P, F)), lo, hi, None),
            ('kinks', vee, np.zeros((P, F)), lo, hi, [np.array([-1., 0., 1.]), np.zeros(0), np.zeros(0)]),
            ('kinks, off-kink start', vee, np.tile([0.4, -2.0, 1.0], (P, 1)), lo, hi, [np.array([-1., 0., 1.]), np.zeros(0), np.zeros(0)])]


@pytest.mark.parametrize('case', range(3))
def test_native_loop_equals_the_numpy_form(case):
    """bi_minimize_batched is the numpy algorithm statement by statement: same iterates, same verdicts, same number of
    evaluation calls (the linear solves differ in their last bits, so results agree to rounding, not bitwise)."""
    from blueice_amd.profile import batched_minimize
    name, fun, x0, lo, hi, kinks = _engine_cases()[case]
    log = {'native': [], 'numpy': []}
    out = {}
    for engine in ('native', 'numpy'):
        def counted(x, rows, engine=engine):
            log[engine].append(len(rows))
            return fun(x, rows)
        out[engine] = batched_minimize(counted, x0, lo, hi, gtol=1e-8, max_iter=300, kinks=kinks, engine=engine)
    (xa, fa, ia), (xb, fb, ib) = out['native'], out['numpy']
    for k in ('converged', 'stalled', 'failed'):
        np.testing.assert_array_equal(ia[k], ib[k], err_msg='%s: %s' % (name, k))
    ok = ~ia['failed']
    np.testing.assert_allclose(fa[ok], fb[ok], rtol=1e-9, atol=1e-12, err_msg=name)
    np.testing.assert_allclose(xa[ok], xb[ok], rtol=1e-6, atol=1e-7, err_msg=name)
    assert ia['calls'] == ib['calls'] == len(log['native']) == len(log['numpy'])
    assert ia['kink_calls'] == ib['kink_calls'] and ia['iterations'] == ib['iterations']
    assert log['native'] == log['numpy'], 'the two forms asked for different batches'


def test_native_loop_carries_exceptions_and_options(d2):
    from blueice_amd.profile import batched_minimize

    def broken(x, rows):
        if len(rows) < 5:
            raise RuntimeError('objective failed')
        return np.sum(x ** 2, axis=1) + np.where(rows % 2 == 0, 0.0, 1.0) * x[:, 0], 2 * x + np.where(rows % 2 == 0, 0.0, 1.0)[:, None] * np.array([1.0, 0.0])

    with pytest.raises(RuntimeError, match='objective failed'):
        batched_minimize(lambda x, rows: broken(x, rows[:3]) if False else (_ for _ in ()).throw(RuntimeError('objective failed')),
                         np.ones((6, 2)), np.full(2, -5.), np.full(2, 5.))
    # tuning options select the numpy form (the C++ loop has the defaults compiled in)
    x, f, info = batched_minimize(broken, np.ones((6, 2)), np.full(2, -5.), np.full(2, 5.), c1=1e-3)
    assert info['converged'].all()
    # the whole engine on the likelihood stand-in, both forms
    lf = OracleLikelihood(d2['model'], d2['counts'], ['shift', 'stretch'], analytic=True)
    from blueice_amd import profile
    res = {}
    for engine in ('native', 'numpy'):
        profile.ENGINE = engine
        try:
            res[engine] = profile.bestfit_batched(lf, points={'shift': np.linspace(-0.8, 0.8, 9)}, stretch=1.0, s2_rate_multiplier=1.0, return_info=True)
        finally:
            profile.ENGINE = 'native'
    np.testing.assert_allclose(res['native'][1], res['numpy'][1], rtol=1e-9)
    assert res['native'][2]['calls'] == res['numpy'][2]['calls']
