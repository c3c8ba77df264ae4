"""Fit drivers over a likelihood callable: the `bestfit_scipy` path of the reference
(blueice/inference.py:57-178) plus a batched `best_anchor` (:34-54).

These only need `lf(**kwargs) -> float`, `lf.rate_parameters`, `lf.shape_parameters`,
`lf.get_bounds`, `lf.pdf_base_config`; scipy.optimize is used as is.  iminuit / emcee drivers and
plotting are out of scope.

Profiled quantities -- `likelihood_ratio_scan` with floating nuisances, `one_parameter_interval` -- run on the batched
profile-fit engine (blueice_amd.profile: all hypotheses advance together, one device call per optimiser iteration)
whenever the likelihood offers batched evaluation; a user-supplied `bestfit_routine` keeps the reference's sequential
loops (blueice/inference.py:332-443).
"""
from collections import OrderedDict
from copy import deepcopy

import numpy as np
from scipy import stats
from scipy.optimize import brentq, minimize

from .exceptions import NoOpimizationNecessary, OptimizationFailed
from .profile import bestfit_batched, supports_batched_fits
from .utils import is_numeric

__all__ = ['best_anchor', 'make_objective', 'bestfit_scipy', 'bestfit_device', 'bestfit_batched', 'bestfit_toys', 'toy_mc_fits',
           'one_parameter_interval', 'likelihood_ratio_scan']


def best_anchor(lf):
    """Shape-parameter dict of the anchor model with the highest likelihood -- all anchors in one
    batched device call when the likelihood offers `eval_points`."""
    if not len(lf.shape_parameters):
        return dict()
    names = list(lf.shape_parameters.keys())
    anchors = list(lf.anchor_models.keys())
    if hasattr(lf, 'eval_points'):
        results = lf.eval_points({n: [a[j] for a in anchors] for j, n in enumerate(names)})
    else:
        results = np.array([lf(**dict(zip(names, a))) for a in anchors])
    return dict(zip(names, anchors[int(np.argmax(results))]))


_FD_STEP = float(np.finfo(np.float64).eps) ** 0.5          # scipy.optimize._numdiff: relative step of '2-point' differences


def make_objective(lf, guess=None, minus=True, rates_in_log_space=False, with_gradient=False,
                   stencil_respects_bounds=False, **kwargs):
    """-> (f(x), names, guesses, bounds) over the parameters not fixed through kwargs.
    Rate multipliers come first (guess 1, bounds (0, None)), then shape parameters (bounds from the
    anchors, guess = base setting).  with_gradient (extensions, both for `scipy.optimize.minimize(..., jac=True)`):
    True -> f returns (value, analytic gradient) from one device pass (`lf.value_and_gradient`);
    'stencil' -> f returns (value, scipy's forward-difference gradient), the F + 1 stencil points evaluated in ONE
    batched device call (`lf.eval_points`)."""
    guess = guess or {}
    names, guesses, bounds = [], [], []
    for src in lf.rate_parameters:
        key = '%s_rate_multiplier' % src
        if key in kwargs:
            continue
        g = guess.get(key, 1)
        names.append(key)
        guesses.append(np.log10(g) if rates_in_log_space else g)
        bounds.append((None, None) if rates_in_log_space else (0, None))
    for key, (_, _, base_value) in lf.shape_parameters.items():
        if key in kwargs:
            continue
        g = guess.get(key)
        if g is None:
            g = lf.pdf_base_config.get(key)
            if not is_numeric(g):
                g = base_value
        names.append(key)
        guesses.append(g)
        bounds.append(lf.get_bounds(key))
    if not names:
        raise NoOpimizationNecessary("There are no parameters to fit, no optimization is necessary")
    sign = -1 if minus else 1
    log_rate = [rates_in_log_space and n.endswith('_rate_multiplier') for n in names]

    def objective(args):
        call = {n: (10 ** a if lg else a) for n, a, lg in zip(names, args, log_rate)}
        call.update(kwargs)
        return lf(**call) * sign

    def objective_with_gradient(args):
        call = {n: (10 ** a if lg else a) for n, a, lg in zip(names, args, log_rate)}
        call.update(kwargs)
        value, grads = lf.value_and_gradient(**call)
        g = np.array([grads[n] * (np.log(10.) * call[n] if lg else 1.0) for n, lg in zip(names, log_rate)])
        if not np.isfinite(value):
            g = np.zeros(len(names))
        return value * sign, g * sign

    def objective_with_stencil(args):
        """Value AND scipy's own forward-difference gradient from ONE batched device call: scipy's default minimiser
        differences the objective numerically, F + 1 scalar calls per gradient (blueice/inference.py:111-124,153-155 via
        scipy.optimize._numdiff); the F + 1 stencil points are independent, so they go to `lf.eval_points` together --
        same points (scipy's step sqrt(eps) * sign(x) * max(1, |x|), turned around at an upper bound), same
        differences, one launch in which the points of a grid cell share one pass over its templates."""
        args = np.asarray(args, dtype=float)
        F = len(args)
        h = _FD_STEP * np.where(args >= 0, 1.0, -1.0) * np.maximum(1.0, np.abs(args))
        for j, (lo_j, hi_j) in enumerate(stencil_bounds):
            if hi_j is not None and args[j] + h[j] > hi_j:
                h[j] = -abs(h[j])
            elif lo_j is not None and args[j] + h[j] < lo_j:
                h[j] = abs(h[j])
        pts = np.repeat(args[None, :], F + 1, axis=0)
        pts[np.arange(1, F + 1), np.arange(F)] += h
        call = {n: (10 ** pts[:, j] if lg else pts[:, j]) for j, (n, lg) in enumerate(zip(names, log_rate))}
        call.update(kwargs)
        vals = np.asarray(lf.eval_points(call), dtype=float) * sign
        with np.errstate(invalid='ignore'):
            grad = (vals[1:] - vals[0]) / ((args + h) - args)
        return vals[0], grad

    # bounds the stencil must respect: only those the minimiser is told about (pass_bounds_to_minimizer); without them
    # scipy steps blindly, and so does this
    stencil_bounds = [(None, None)] * len(names) if not stencil_respects_bounds else \
        [(None if b[0] in (None, -np.inf) else b[0], None if b[1] in (None, np.inf) else b[1]) for b in bounds]
    if with_gradient == 'stencil':
        return objective_with_stencil, names, np.array(guesses), bounds
    return (objective_with_gradient if with_gradient else objective), names, np.array(guesses), bounds


def bestfit_scipy(lf, minimize_kwargs=None, rates_in_log_space=False, pass_bounds_to_minimizer=False,
                  use_gradient=False, batch_stencil=True, **kwargs):
    """Maximise lf over its floating parameters -> (OrderedDict name -> value, max log likelihood).
    scipy's default minimizer first, Nelder-Mead as the fallback, OptimizationFailed after that.
    use_gradient=True (extension): hand scipy the analytic gradient computed in the same device pass as
    the value instead of letting it difference the objective numerically (n_parameters + 1 calls per step).
    batch_stencil (default on, when the likelihood evaluates batches): scipy still gets its own forward differences,
    but the n_parameters + 1 points behind each of them are evaluated in one device call (`make_objective`,
    with_gradient='stencil'); batch_stencil=False is the reference's stream of scalar calls."""
    minimize_kwargs = minimize_kwargs or {}
    use_gradient = use_gradient and bool(getattr(lf, 'supports_gradient', False))
    # scipy's gradient-based default methods difference the objective numerically: hand them the same differences, with
    # the stencil evaluated as one batch (not for methods that take no gradient, nor when the caller brings a jac)
    stencil = batch_stencil and not use_gradient and hasattr(lf, 'eval_points') and 'jac' not in minimize_kwargs and \
        str(minimize_kwargs.get('method', 'BFGS')).lower() in ('bfgs', 'l-bfgs-b', 'cg', 'slsqp', 'tnc')
    mode = dict(with_gradient=True) if use_gradient else \
        (dict(with_gradient='stencil', stencil_respects_bounds=pass_bounds_to_minimizer) if stencil else {})
    try:
        f, names, guess, bounds = lf.make_objective(minus=True, rates_in_log_space=rates_in_log_space, **dict(kwargs, **mode))
    except NoOpimizationNecessary:
        return {}, lf(**kwargs)
    use_bounds = bounds if pass_bounds_to_minimizer else None
    res = minimize(f, guess, bounds=use_bounds, **(dict(minimize_kwargs, jac=True) if mode else minimize_kwargs))
    if not res.success:
        retry = deepcopy(minimize_kwargs)
        retry.pop('method', None)
        if mode:
            f, names, guess, bounds = lf.make_objective(minus=True, rates_in_log_space=rates_in_log_space, **kwargs)
        res = minimize(f, guess, bounds=use_bounds, method='Nelder-Mead', **retry)
        if not res.success:
            raise OptimizationFailed("Optimization failure: ", res)
    x = res.x if len(names) != 1 else [res.x.item()]
    out = OrderedDict()
    for n, v in zip(names, x):
        out[n] = 10 ** v if (rates_in_log_space and n.endswith('_rate_multiplier')) else v
    return out, -res.fun


def bestfit_device(lf, guess=None, **kwargs):
    """`bestfit_scipy`'s signature and return value -- (OrderedDict name -> float, max log likelihood) -- from the batched
    engine with a single problem: analytic gradient, the kinks of the morph at the anchors handled as kinks (every fit
    starts on one: base values are anchors), starts in the other grid cells.  On C2 about half the time of scipy's
    minimiser on the same device likelihood, and a maximum that is never lower.  A drop-in wherever the reference takes a
    `bestfit_routine` (blueice/inference.py:324-330).  kwargs: parameters held fixed, as `bestfit_scipy`."""
    if not supports_batched_fits(lf):
        return bestfit_scipy(lf, guess=guess, **kwargs)
    try:
        best, ll = bestfit_batched(lf, guess=guess, **kwargs)
    except NoOpimizationNecessary:
        return {}, lf(**kwargs)
    return OrderedDict((k, float(v[0])) for k, v in best.items()), float(ll[0])


def bestfit_toys(lf, t0=0, t1=None, **kwargs):
    """Fit every dataset the likelihood holds -- the toys of `simulate_toys`, or a stack handed to `set_binned_data` --
    at the same time: one problem per dataset on the batched profile-fit engine, one device call per optimiser iteration
    for all of them.  The reference's toy-MC loop is `d = lf.base_model.simulate(); lf.set_data(d); bestfit_scipy(lf)`,
    one toy after the other (blueice/model.py:69-91, inference.py:131-178).  kwargs: as `bestfit_batched` (fixed
    parameters, guess, ...).  -> (OrderedDict name -> fitted values [t1 - t0], max log likelihood [t1 - t0]).

    Device-generated toys exist as non-empty-bin lists only; evaluating them at a parameter point of their own needs the
    compacted templates of every toy (rows x non-empty bins x 8 bytes per toy: 41 MB at 4 sources x 5^3 anchors and
    10^4 events), within `compact_budget` (16 GB unless raised with lf.ctx.set_param BEFORE the toys are made)."""
    ctx = getattr(lf, 'ctx', None)
    if ctx is None:
        raise NotImplementedError("bestfit_toys needs a likelihood with its datasets on one device context")
    t1 = ctx.T if t1 is None else t1
    if not 0 <= t0 < t1 <= ctx.T:
        raise ValueError("datasets [%d, %d) of %d" % (t0, t1, ctx.T))
    return bestfit_batched(lf, datasets=np.arange(t0, t1), **kwargs)


def toy_mc_fits(lf, n_toys, chunk=256, seed=0, truth=None, livetime_days=None, first_toy=0, **fit_kwargs):
    """A toy-MC ensemble with a fit per toy, start to finish on the device: `n_toys` binned toys drawn at the parameter
    values `truth` (dict; defaults elsewhere) and fitted, `chunk` toys at a time (`simulate_toys` + `bestfit_toys`) -- the
    reference's `for _ in range(n_toys): d = lf.base_model.simulate(); lf.set_data(d); bestfit_scipy(lf)`
    (blueice/model.py:69-91, inference.py:131-178).  The toys are numbered globally (the generator's counters are
    (seed, toy number, bin)), so the ensemble does not depend on `chunk`: that only bounds the HBM taken by the toys'
    compacted templates (see `bestfit_toys`).  first_toy: the number of this call's first toy -- ranks that each take a range
    of one ensemble (one process per GPU) draw the toys one process would.  fit_kwargs: parameters held fixed, `guess`, ...
    as `bestfit_batched`.
    -> (OrderedDict name -> fitted values [n_toys], max log likelihood [n_toys]).  Afterwards the likelihood's data are
    the toys of the last chunk."""
    ctx = getattr(lf, 'ctx', None)
    if ctx is None or not hasattr(lf, 'simulate_toys'):
        raise NotImplementedError("toy_mc_fits needs a binned likelihood on one device context")
    best, lls = None, []
    try:
        for t0 in range(0, int(n_toys), int(chunk)):
            n = min(int(chunk), int(n_toys) - t0)
            ctx.set_param('toy_offset', int(first_toy) + t0)
            lf.simulate_toys(n, seed=seed, livetime_days=livetime_days, **(truth or {}))
            b, ll = bestfit_toys(lf, livetime_days=livetime_days, **fit_kwargs)
            lls.append(ll)
            if best is None:
                best = OrderedDict((k, [v]) for k, v in b.items())
            else:
                for k, v in b.items():
                    best[k].append(v)
    finally:
        ctx.set_param('toy_offset', 0)
    return OrderedDict((k, np.concatenate(v)) for k, v in best.items()), np.concatenate(lls)


def _first_crossing(tfun, a, b, xtol=1e-11, points_per_round=16, max_rounds=12):
    """The root of t between a and b that lies nearest to a, by rounds of batched evaluations: every round evaluates a
    fan of hypotheses inside the current bracket in ONE call of tfun(h [n]) -> t [n] -- uniformly spaced at first, then
    half of them clustered around the secant estimate (t is smooth) -- and keeps the first sign change seen from a.
    Like brentq, raises ValueError when t(a) and t(b) have the same sign."""
    ta, tb = tfun(np.array([a, b], dtype=float))
    if ta == 0:
        return float(a)
    if tb == 0:
        return float(b)
    if not np.isfinite(ta) or not np.isfinite(tb) or np.sign(ta) == np.sign(tb):
        raise ValueError("f(a) and f(b) must have different signs")
    xs, ts = np.array([a, b], dtype=float), np.array([ta, tb], dtype=float)
    K = points_per_round
    for rnd in range(max_rounds):
        (lo, hi), (tl, th) = xs, ts
        if abs(hi - lo) <= xtol * max(1.0, abs(lo), abs(hi)):
            break
        fan = np.linspace(lo, hi, K + 2)[1:-1]
        if rnd >= 1:
            secant = lo - tl * (hi - lo) / (th - tl)
            steps = abs(hi - lo) * 0.5 ** np.arange(3, 3 + 3 * (K // 4), 3)
            near = secant + np.concatenate([-steps, [0.0], steps])
            near = near[(near > min(lo, hi)) & (near < max(lo, hi))]
            fan = np.unique(np.concatenate([np.linspace(lo, hi, K // 2 + 2)[1:-1], near]))
            if hi < lo:
                fan = fan[::-1]
        allx = np.concatenate([[lo], fan, [hi]])
        allt = np.concatenate([[tl], tfun(fan), [th]])
        k = np.flatnonzero(np.sign(allt[1:]) != np.sign(allt[0]))[0]          # first sign change seen from a
        xs, ts = allx[k:k + 2], allt[k:k + 2]
        if ts[1] == 0:
            return float(xs[1])
    (lo, hi), (tl, th) = xs, ts
    return float(lo - tl * (hi - lo) / (th - tl))


# options of bestfit_scipy that the reference's callers pass through `**kwargs` of the scan / interval drivers
# (blueice/inference.py:332-443 forward them to the fit routine) -- they are NOT fixed parameters
_FIT_ROUTINE_OPTIONS = ('minimize_kwargs', 'rates_in_log_space', 'pass_bounds_to_minimizer', 'use_gradient', 'batch_stencil',
                        'guess')


def _takes_fit_routine_options(kwargs):
    """True when the caller handed options of the fit routine along with the fixed parameters: those calls keep the
    reference's sequential loop over `bestfit_scipy(lf, **kwargs)`, which understands them."""
    return any(k in kwargs for k in _FIT_ROUTINE_OPTIONS)


def one_parameter_interval(lf, target, bound, confidence_level=0.9, kind='upper', bestfit_routine=None, fit_options=None,
                           t_ppf=None, **kwargs):
    """Profile-likelihood interval on parameter `target` (reference: blueice/inference.py:332-389).
    kind 'upper' / 'lower': `bound` is the far end of the line search; 'central': a 2-tuple.
    The test statistic 2 (max logL - logL profiled at the hypothesis) is compared with
    norm.ppf(quantile)**2 (Wilks) or with t_ppf(hypothesis, quantile); the crossing is the one the reference's brentq
    search finds.  With a likelihood that evaluates batches (and no bestfit_routine of the caller's) the search runs on
    the batched profile-fit engine: every round profiles a fan of hypotheses in lock-step on the device (a handful of
    rounds of ~16 fits, each a few dozen device calls) instead of brentq's chain of nested sequential fits (3 387 scalar
    likelihood calls per limit in SURVEY.md's probe); otherwise the reference's loop.  fit_options: dict of options of the
    batched engine (`bestfit_batched`: multi_start='cells', gtol, ...)."""
    fit_options = dict(fit_options or {})
    if target is None:
        target = lf.source_list[-1] + '_rate_multiplier'
    # (options of bestfit_scipy among the kwargs -- minimize_kwargs, pass_bounds_to_minimizer, ... -- go where the
    # reference sends them: to the fit routine, point by point)
    batched = bestfit_routine is None and supports_batched_fits(lf) and not _takes_fit_routine_options(kwargs)
    fit = bestfit_routine or bestfit_scipy
    if batched:
        try:
            best, ll = bestfit_batched(lf, **fit_options, **kwargs)
            best, max_ll = {k: float(v[0]) for k, v in best.items()}, float(ll[0])
        except NoOpimizationNecessary:
            batched = False
    if not batched:
        best, max_ll = fit(lf, **kwargs)
    global_best = best[target]

    def critical_of(hypothesis, quantile):
        return stats.norm.ppf(quantile) ** 2 if t_ppf is None else t_ppf(hypothesis, quantile)

    def one_sided_ok(hypothesis):
        return (kind == 'upper' and hypothesis <= global_best) or (kind == 'lower' and hypothesis >= global_best)

    def t(hypothesis, quantile):
        critical = critical_of(hypothesis, quantile)
        if one_sided_ok(hypothesis):
            return 0 - critical
        _, ll = fit(lf, **dict(kwargs, **{target: hypothesis}))
        return 2 * (max_ll - ll) - critical

    def t_batched(quantile):
        nuisances = [k for k in best if k != target]

        def tfun(hs):
            hs = np.asarray(hs, dtype=float)
            crit = np.array([critical_of(h, quantile) for h in hs])
            out = 0.0 - crit
            need = np.array([not one_sided_ok(h) for h in hs])
            if np.any(need):
                if nuisances:                       # every hypothesis starts where the reference starts AND at the global best fit's nuisances
                    _, ll = bestfit_batched(lf, points={target: hs[need]}, also_from=[{k: best[k] for k in nuisances}], **fit_options, **kwargs)
                else:                               # nothing left to profile: plain evaluations
                    ll = np.asarray(lf.eval_points(dict(kwargs, **{target: hs[need]})))
                out[need] = 2 * (max_ll - ll) - crit[need]
            return out
        return tfun

    def search(a, b, quantile):
        if batched:
            return _first_crossing(t_batched(quantile), a, b)
        return brentq(t, a, b, args=(quantile,))

    if kind == 'central':
        return (search(bound[0], global_best, (1 - confidence_level) / 2),
                search(global_best, bound[1], 1 - (1 - confidence_level) / 2))
    if kind == 'lower':
        return search(bound, global_best, 1 - confidence_level)
    if kind == 'upper':
        return search(global_best, bound, confidence_level)
    raise ValueError("kind must be 'upper', 'lower' or 'central'")


def likelihood_ratio_scan(lf, *space, bestfit_routine=None, fit_options=None, **kwargs):
    """-log likelihood ratio over a 1-d or 2-d grid of parameter values: the numbers behind the reference's
    `plot_likelihood_ratio` (blueice/inference.py:392-443) without the plotting.
    space: (name, values) tuples.  Parameters given in kwargs are fixed, all others are fitted at every grid
    point.  When nothing is left to fit the whole grid is ONE batched device call (`lf.eval_points`); with floating
    nuisances the grid points are profiled together on the batched fit engine (blueice_amd.profile: one device call per
    optimiser iteration over the whole grid) -- unless the caller brings a `bestfit_routine`, which is then run point by
    point as the reference does.  Returns an array of shape [len(values_0)(, len(values_1))], best point = 0."""
    if not 1 <= len(space) <= 2:
        raise ValueError("Can't handle %d dimensions" % len(space))
    names = [n for n, _ in space]
    grids = np.meshgrid(*[np.asarray(v, dtype=float) for _, v in space], indexing='ij')
    floating = [p + '_rate_multiplier' for p in lf.rate_parameters if p + '_rate_multiplier' not in kwargs] + \
               [p for p in lf.shape_parameters if p not in kwargs]
    floating = [p for p in floating if p not in names]
    pts = {n: g.ravel() for n, g in zip(names, grids)}
    routine_options = _takes_fit_routine_options(kwargs)
    if not floating and hasattr(lf, 'eval_points'):
        pts.update({k: v for k, v in kwargs.items() if k not in _FIT_ROUTINE_OPTIONS})
        ll = np.asarray(lf.eval_points(pts)).reshape(grids[0].shape)
    elif floating and bestfit_routine is None and supports_batched_fits(lf) and not routine_options:
        ll = bestfit_batched(lf, points=pts, **(fit_options or {}), **kwargs)[1].reshape(grids[0].shape)
    else:
        fit = bestfit_routine or bestfit_scipy
        ll = np.empty(grids[0].shape)
        for idx in np.ndindex(*grids[0].shape):
            ll[idx] = fit(lf, **dict(kwargs, **{n: float(g[idx]) for n, g in zip(names, grids)}))[1]
    return np.nanmax(ll) - ll
