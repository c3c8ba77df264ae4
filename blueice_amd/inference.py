"""Fit drivers over a likelihood callable: the `bestfit_scipy` path of the reference
(blueice/inference.py:57-178) plus a batched `best_anchor` (:34-54).

These only need `lf(**kwargs) -> float`, `lf.rate_parameters`, `lf.shape_parameters`,
`lf.get_bounds`, `lf.pdf_base_config`; scipy.optimize is used as is.  iminuit / emcee drivers,
intervals and plotting are out of scope.
"""
from collections import OrderedDict
from copy import deepcopy

import numpy as np
from scipy import stats
from scipy.optimize import brentq, minimize

from .exceptions import NoOpimizationNecessary, OptimizationFailed
from .utils import is_numeric

__all__ = ['best_anchor', 'make_objective', 'bestfit_scipy', 'one_parameter_interval', 'likelihood_ratio_scan']


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


def make_objective(lf, guess=None, minus=True, rates_in_log_space=False, with_gradient=False, **kwargs):
    """-> (f(x), names, guesses, bounds) over the parameters not fixed through kwargs.
    Rate multipliers come first (guess 1, bounds (0, None)), then shape parameters (bounds from the
    anchors, guess = base setting).  with_gradient=True (extension): f returns (value, gradient) from one
    device pass (`lf.value_and_gradient`), for `scipy.optimize.minimize(..., jac=True)`."""
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

    return (objective_with_gradient if with_gradient else objective), names, np.array(guesses), bounds


def bestfit_scipy(lf, minimize_kwargs=None, rates_in_log_space=False, pass_bounds_to_minimizer=False,
                  use_gradient=False, **kwargs):
    """Maximise lf over its floating parameters -> (OrderedDict name -> value, max log likelihood).
    scipy's default minimizer first, Nelder-Mead as the fallback, OptimizationFailed after that.
    use_gradient=True (extension): hand scipy the analytic gradient computed in the same device pass as
    the value instead of letting it difference the objective numerically (n_parameters + 1 calls per step)."""
    minimize_kwargs = minimize_kwargs or {}
    use_gradient = use_gradient and bool(getattr(lf, 'supports_gradient', False))
    try:
        f, names, guess, bounds = lf.make_objective(minus=True, rates_in_log_space=rates_in_log_space,
                                                    **(dict(kwargs, with_gradient=True) if use_gradient else kwargs))
    except NoOpimizationNecessary:
        return {}, lf(**kwargs)
    use_bounds = bounds if pass_bounds_to_minimizer else None
    res = minimize(f, guess, bounds=use_bounds, **(dict(minimize_kwargs, jac=True) if use_gradient else minimize_kwargs))
    if not res.success:
        retry = deepcopy(minimize_kwargs)
        retry.pop('method', None)
        if use_gradient:
            f, names, guess, bounds = lf.make_objective(minus=True, rates_in_log_space=rates_in_log_space, **kwargs)
        res = minimize(f, guess, bounds=use_bounds, method='Nelder-Mead', **retry)
        if not res.success:
            raise OptimizationFailed("Optimization failure: ", res)
    x = res.x if len(names) != 1 else [res.x.item()]
    out = OrderedDict()
    for n, v in zip(names, x):
        out[n] = 10 ** v if (rates_in_log_space and n.endswith('_rate_multiplier')) else v
    return out, -res.fun


def one_parameter_interval(lf, target, bound, confidence_level=0.9, kind='upper', bestfit_routine=None,
                           t_ppf=None, **kwargs):
    """Profile-likelihood interval on parameter `target` (reference: blueice/inference.py:332-389).
    kind 'upper' / 'lower': `bound` is the far end of the line search; 'central': a 2-tuple.
    The test statistic 2 (max logL - logL profiled at the hypothesis) is compared with
    norm.ppf(quantile)**2 (Wilks) or with t_ppf(hypothesis, quantile); the crossing is found with brentq.
    Every profile point is one nested fit, i.e. a stream of single-point device calls."""
    fit = bestfit_routine or bestfit_scipy
    if target is None:
        target = lf.source_list[-1] + '_rate_multiplier'
    best, max_ll = fit(lf, **kwargs)
    global_best = best[target]

    def t(hypothesis, quantile):
        critical = stats.norm.ppf(quantile) ** 2 if t_ppf is None else t_ppf(hypothesis, quantile)
        one_sided_ok = (kind == 'upper' and hypothesis <= global_best) or (kind == 'lower' and hypothesis >= global_best)
        if one_sided_ok:
            return 0 - critical
        _, ll = fit(lf, **dict(kwargs, **{target: hypothesis}))
        return 2 * (max_ll - ll) - critical

    if kind == 'central':
        return (brentq(t, bound[0], global_best, args=((1 - confidence_level) / 2,)),
                brentq(t, global_best, bound[1], args=(1 - (1 - confidence_level) / 2,)))
    if kind == 'lower':
        return brentq(t, bound, global_best, args=(1 - confidence_level,))
    if kind == 'upper':
        return brentq(t, global_best, bound, args=(confidence_level,))
    raise ValueError("kind must be 'upper', 'lower' or 'central'")


def likelihood_ratio_scan(lf, *space, bestfit_routine=None, **kwargs):
    """-log likelihood ratio over a 1-d or 2-d grid of parameter values: the numbers behind the reference's
    `plot_likelihood_ratio` (blueice/inference.py:392-443) without the plotting.
    space: (name, values) tuples.  Parameters given in kwargs are fixed, all others are fitted at every grid
    point; when nothing is left to fit the whole grid is ONE batched device call (`lf.eval_points`).
    Returns an array of shape [len(values_0)(, len(values_1))], best point = 0."""
    if not 1 <= len(space) <= 2:
        raise ValueError("Can't handle %d dimensions" % len(space))
    fit = bestfit_routine or bestfit_scipy
    names = [n for n, _ in space]
    grids = np.meshgrid(*[np.asarray(v, dtype=float) for _, v in space], indexing='ij')
    floating = [p + '_rate_multiplier' for p in lf.rate_parameters if p + '_rate_multiplier' not in kwargs] + \
               [p for p in lf.shape_parameters if p not in kwargs]
    floating = [p for p in floating if p not in names]
    if not floating and hasattr(lf, 'eval_points'):
        pts = {n: g.ravel() for n, g in zip(names, grids)}
        pts.update({k: v for k, v in kwargs.items()})
        ll = np.asarray(lf.eval_points(pts)).reshape(grids[0].shape)
    else:
        ll = np.empty(grids[0].shape)
        for idx in np.ndindex(*grids[0].shape):
            ll[idx] = fit(lf, **dict(kwargs, **{n: float(g[idx]) for n, g in zip(names, grids)}))[1]
    return np.nanmax(ll) - ll
